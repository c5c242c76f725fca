// Conv weight gradient on PRE-SPLIT operands for gfx950: dW[k][tap][c] = sum over output pixels of dY[pix][k] * X[pix @ tap][c].
//
// conv_igemm.hip's wgrad kernels split both f32 operands into three bf16 parts while staging them -- every dy / x slab is
// converted again by each of the up to 18 workgroups (taps x tiles) that read it.  Here both operands arrive as x3 rows
// (csrc/conv_x3.hip): dy3 from the BN-backward pass, x3 from the forward's producer, so the loop is copy + MFMA only:
//   * the reduction index is the PIXEL: a 32-pixel slab of 32 channels of one part is two 1 KiB pieces
//     [16 pixels][32 ch] bf16, fetched by ONE LDS-DMA instruction each (lane = pixel l >> 2, 16-B chunk l & 3);
//     out-of-image taps and pixels beyond the split read the operand's zero row;
//   * MFMA fragments want 8 consecutive pixels of one channel per lane while the image is [pixel][channel]: gfx950's
//     transposed LDS read (ds_read_b64_tr_b16) delivers exactly that, conflict-free on unpadded 64-B rows;
//   * same ring / one barrier per slab / counted vmcnt / half-slab skew as conv_x3_kernel;
//   * one workgroup = (K-tile, C-tile, tap, pixel split); partial tiles are added into dW with f32 atomics in
//     two-128-B-segment wave instructions (the shape that runs at the full atomic rate).
// Reference site replaced: the weight gradient autograd derives for every dense nn.Conv2d of the DeepLab path.
#include "dass_common.h"
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>

namespace {

struct WX3P {
    const char *dy3;   // [M + 1][KC][192], row M zero
    const char *x3;    // [rows_in + 1][CC][192], row rows_in zero
    float *dw;         // [K][R*S][C] f32, accumulated atomically
    unsigned dy3_bytes, x3_bytes, dy_zero, x_zero, dy_pitch, x_pitch;
    unsigned dy_tr, x_tr;  // trailers {inv_scale, bound} of the two-part format
    int N, H, W, C, OH, OW, K, R, S, stride, pad, dil;
    int M, KC, CC, ktiles, ctiles, psplit, pix_per_split;
    unsigned mg_ohw, mg_ow;  // magic multipliers of the divisions by OH * OW and by OW (dass_common.h x3_fastdiv), filled by wx3_finish
    int sh_ohw, sh_ow;
};

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

// FOUR LDS-DMA instructions into four CONSECUTIVE 1 KiB pieces with ONE write of M0: the instruction's immediate offset is added
// to the LDS address (M0 + offset + 16 * lane) as well as to the memory address (base + voffset + soffset + offset), so piece i goes
// out with offset:1024 i and a per-lane voffset that carries its true source minus 1024 i.  soff: wave-uniform part of the source
// offset (the 32-channel chunk; + WX3_BIAS, see wgrad_x3_body).  M0 is not restored: nothing the compiler emits for these kernels
// reads it (gfx9+ LDS instructions do not), and every statement that needs it writes it itself.
// (Round 3 issued every piece as `save m0; set m0; nop; load; restore m0` -- 40 of the loop's ~280 SALU instructions per slab.)
__device__ __forceinline__ void wdma16x4(v4i rsrc, unsigned lds, unsigned soff, unsigned v0, unsigned v1, unsigned v2, unsigned v3) {
    asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %0, %5, %6 offen lds\n\t"
                 "buffer_load_dwordx4 %1, %5, %6 offen offset:1024 lds\n\t"
                 "buffer_load_dwordx4 %2, %5, %6 offen offset:2048 lds\n\t"
                 "buffer_load_dwordx4 %3, %5, %6 offen offset:3072 lds"
                 :
                 : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(lds), "s"(rsrc), "s"(soff)
                 : "memory");
}
__device__ __forceinline__ void wdma16x6(v4i rsrc, unsigned lds, unsigned soff, unsigned v0, unsigned v1, unsigned v2, unsigned v3, unsigned v4,
                                         unsigned v5) {  // three-part rows: six pieces per chunk, the last two behind a second M0
    wdma16x4(rsrc, lds, soff, v0, v1, v2, v3);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %0, %3, %4 offen lds\n\t"
                 "buffer_load_dwordx4 %1, %3, %4 offen offset:1024 lds"
                 :
                 : "v"(v4), "v"(v5), "s"(lds + 4096u), "s"(rsrc), "s"(soff)
                 : "memory");
}
__device__ __forceinline__ void wdma16x2(v4i rsrc, unsigned lds, unsigned soff, unsigned v0, unsigned v1) {  // one-part rows: two pieces per chunk
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %0, %3, %4 offen lds\n\t"
                 "buffer_load_dwordx4 %1, %3, %4 offen offset:1024 lds"
                 :
                 : "v"(v0), "v"(v1), "s"(lds), "s"(rsrc), "s"(soff)
                 : "memory");
}
constexpr unsigned WX3_BIAS = 8192u;  // the buffer descriptors start this many bytes BELOW the operands, so that `source - 1024 i` never wraps
__device__ __forceinline__ v4i wmake_srd(const void *base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    v4i r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
template <int N> __device__ __forceinline__ void wwait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// transposed LDS read (ds_read_b64_tr_b16) through the builtin: the compiler folds the piece offsets into the instruction's
// offset field, lands the two reads of a 16-byte MFMA operand directly in the halves of its register tuple and places the
// lgkmcnt waits in front of the first use.  (The asm form this replaces cost a v_add per read and two v_mov_b64 per operand:
// 662 instructions per 32-pixel slab around 24 MFMAs -- the loop was issue-bound, 31 % MFMA-busy.  The LDS-DMA stays in asm,
// so the compiler has no reason to drain vmcnt in front of these reads.)
typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
__device__ __forceinline__ v4s tr16(unsigned addr) {
    typedef __attribute__((address_space(3))) v4s *lds_ptr;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(size_t)addr);
}

// Tile BMK (out channels) x BNC (in channels); NW waves as WARPS_M x WARPS_N.  LDS stage = 32 pixels:
// operand A (dy): BMK/32 chunks x 3 parts x 2 pixel halves, piece (chunk ch, part pl, half hf) at ((ch*3 + pl)*2 + hf) * 1024;
// operand B (x) behind it with BNC/32 chunks.
// NP: parts per element (dass_common.h "x3 operand formats"): 3 = bf16 triple / six products, 2 = scaled f16 pair / three products
template <int BMK, int BNC, int WARPS_M, int WARPS_N, int NSTAGE, int NP>
__device__ __forceinline__ void wgrad_x3_body(const WX3P &p, int wg, char *smem) {
    constexpr int NW = WARPS_M * WARPS_N;
    constexpr int SB = NP * 64, CHB = NP * 2048;  // bytes of a row-slab in memory / of one 32-channel chunk of a 32-pixel LDS stage
    constexpr int NTERM = NP == 3 ? 6 : NP == 2 ? 3 : 1;
    constexpr int TMW = BMK / WARPS_M, TNW = BNC / WARPS_N, MT = TMW / 32, NT = TNW / 32;
    constexpr int ACH = BMK / 32, BCH = BNC / 32;           // 32-channel chunks per operand
    constexpr int A_BYTES = ACH * CHB, B_BYTES = BCH * CHB, STAGE = A_BYTES + B_BYTES;
    constexpr int CHW = (ACH + BCH) / NW;                   // chunks a wave loads (all three parts, both pixel halves)
    static_assert((ACH + BCH) % NW == 0 && CHW >= 1, "chunks must divide over the waves");
    static_assert(ACH % CHW == 0, "a wave loads chunks of ONE operand");
    constexpr int G = CHW * NP * 2;                         // DMA instructions per wave and slab
    static_assert(MT >= 1 && NT >= 1 && NSTAGE >= 2 && NSTAGE <= 3 && G * (NSTAGE - 2) <= 63, "tile");
    const int tap = wg % (p.R * p.S);
    wg /= (p.R * p.S);
    const int ct = wg % p.ctiles;
    wg /= p.ctiles;
    const int kt = wg % p.ktiles;
    const int ps = wg / p.ktiles;
    const int k0 = kt * BMK, c0 = ct * BNC;
    const int r = tap / p.S, s2 = tap - r * p.S;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WARPS_N, wn = wave - wm * WARPS_N;
    const int ohw = p.OH * p.OW;
    const long pbeg = (long)ps * p.pix_per_split;
    long pend = pbeg + p.pix_per_split;
    if (pend > p.M) pend = p.M;
    const int total = (int)((pend - pbeg + 31) / 32);       // 32-pixel slabs

    const v4i rsa = wmake_srd(p.dy3 - WX3_BIAS, p.dy3_bytes + WX3_BIAS), rsb = wmake_srd(p.x3 - WX3_BIAS, p.x3_bytes + WX3_BIAS);
    const unsigned smem_base = (unsigned)(size_t)smem;

    // ---- loader role of this wave: chunks first_ch .. first_ch + CHW - 1 of operand A (waves below ACH / CHW) or B.  The NP * 2
    // pieces of one chunk -- (part pl, pixel half hf) at ((ch * NP + pl) * 2 + hf) * 1024 -- are consecutive in LDS: one M0 write
    // per four of them (wdma16x4).  A chunk beyond the tensor (the last tile of K or C) is CLAMPED to the last real chunk, not
    // zeroed: the rows / columns it feeds are never stored.
    const bool loads_a = wave * CHW < ACH;                  // wave-uniform
    const int first_ch = loads_a ? wave * CHW : wave * CHW - ACH;
    const int lp = lane >> 2;                               // pixel row of the piece this lane fetches
    const unsigned chunk16 = (unsigned)((lane & 3) * 16);
    unsigned soff[CHW], ldsoff[CHW];
#pragma unroll
    for (int j = 0; j < CHW; ++j) {
        const int ch = first_ch + j;
        int gch = (loads_a ? k0 : c0) / 32 + ch;
        const int lim = (loads_a ? p.KC : p.CC) - 1;
        gch = gch < lim ? gch : lim;
        soff[j] = __builtin_amdgcn_readfirstlane((unsigned)(gch * SB) + WX3_BIAS);
        ldsoff[j] = __builtin_amdgcn_readfirstlane((unsigned)((loads_a ? 0 : A_BYTES) + ch * CHB));
    }
    // per-lane constant of piece (pl, hf): + pl * 64 + chunk16 (where the 16 bytes sit in the row-slab) - 1024 * (pl * 2 + hf)
    // (what the instruction's immediate offset adds back); pieces 4, 5 of a three-part chunk sit behind the second M0: - 1024 * (i - 4)
    auto piece_k = [&](int pl, int hf) -> unsigned {
        const int i = pl * 2 + hf;
        return (unsigned)(pl * 64) + chunk16 - (unsigned)(1024 * (i < 4 ? i : i - 4));
    };
    const int pend32 = (int)pend;
    const int pix_lane = (int)pbeg + lp;                    // pixel of (slab 0, half 0); slab s, half hf: + 32 s + 16 hf
    const unsigned pitch = loads_a ? p.dy_pitch : p.x_pitch, zero_off = loads_a ? p.dy_zero : p.x_zero;
    const int tap_dy = -p.pad + r * p.dil, tap_dx = -p.pad + s2 * p.dil;
    // operand B source row of output pixel `pix`: the tap's input pixel (1 x 1 / stride 1 / no padding -- two thirds of the layers --:
    // the pixel itself), the zero row when that falls outside the image.  Branch-free, by magic division (no cursor state).
    const bool same_pix = p.R * p.S == 1 && p.pad == 0 && p.stride == 1 && p.H == p.OH && p.W == p.OW;
    auto src_of = [&](int pix, auto ROLE) __attribute__((always_inline)) -> unsigned {
        constexpr int role = decltype(ROLE)::value;         // 0: operand A, 1: B with the pixel itself, 2: B through the tap geometry
        const bool live = pix < pend32;
        if constexpr (role < 2) {
            return live ? (unsigned)pix * pitch : zero_off;
        } else {
            const int pp2 = live ? pix : 0;
            const int n = x3_fastdiv(pp2, p.mg_ohw, p.sh_ohw);
            const int rem = pp2 - n * ohw;
            const int oh = x3_fastdiv(rem, p.mg_ow, p.sh_ow);
            const int ow = rem - oh * p.OW;
            const int iy = oh * p.stride + tap_dy, ix = ow * p.stride + tap_dx;
            const bool in = live && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            return in ? (unsigned)((n * p.H + iy) * p.W + ix) * pitch : zero_off;
        }
    };
    int next_slab = 0;                                       // slab index the next issue fetches
    auto issue_slab = [&](int stage, auto ROLE) __attribute__((always_inline)) {
        const unsigned st = smem_base + (unsigned)(stage * STAGE);
        const int pix = pix_lane + 32 * next_slab;
        ++next_slab;
        const unsigned s0 = src_of(pix, ROLE), s1 = src_of(pix + 16, ROLE);
        const v4i rs = decltype(ROLE)::value == 0 ? rsa : rsb;
        if constexpr (NP == 1) {
            const unsigned v0 = s0 + piece_k(0, 0), v1 = s1 + piece_k(0, 1);
#pragma unroll
            for (int j = 0; j < CHW; ++j) wdma16x2(rs, __builtin_amdgcn_readfirstlane(st + ldsoff[j]), soff[j], v0, v1);
        } else if constexpr (NP == 2) {
            const unsigned v0 = s0 + piece_k(0, 0), v1 = s1 + piece_k(0, 1), v2 = s0 + piece_k(1, 0), v3 = s1 + piece_k(1, 1);
#pragma unroll
            for (int j = 0; j < CHW; ++j) wdma16x4(rs, __builtin_amdgcn_readfirstlane(st + ldsoff[j]), soff[j], v0, v1, v2, v3);
        } else {
            const unsigned v0 = s0 + piece_k(0, 0), v1 = s1 + piece_k(0, 1), v2 = s0 + piece_k(1, 0), v3 = s1 + piece_k(1, 1);
            const unsigned v4 = s0 + piece_k(2, 0), v5 = s1 + piece_k(2, 1);
#pragma unroll
            for (int j = 0; j < CHW; ++j) wdma16x6(rs, __builtin_amdgcn_readfirstlane(st + ldsoff[j]), soff[j], v0, v1, v2, v3, v4, v5);
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- transposed-read lane roles: group g = lane >> 4 -> channel half (g & 1), pixel half of the 16-deep k-step
    // (g >> 1); inside the group lane 4 q + pp supplies the address of pixel row q, 4-channel column chunk pp
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const unsigned lane_off = (unsigned)((8 * (g >> 1) + q) * 64 + (16 * (g & 1) + 4 * pp) * 2);
    const unsigned a_rd = smem_base + (unsigned)((wm * (TMW / 32)) * CHB) + lane_off;
    const unsigned b_rd = smem_base + (unsigned)(A_BYTES + (wn * (TNW / 32)) * CHB) + lane_off;

    // fragments of one 16-pixel k-step: [part][block][pixels 0-3 / 4-7 of this lane's half]; the two 8-byte reads of an MFMA
    // operand land in the halves of its register tuple (tr16 above), the compiler waits for them in front of their first use.
    v4s a0[NP][MT][2], b0[NP][NT][2], a1[NP][MT][2], b1[NP][NT][2];
    auto load_frags = [&](unsigned abase, unsigned bbase, v4s(&a)[NP][MT][2], v4s(&b)[NP][NT][2]) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                b[pl][nt][0] = tr16(bbase + nt * CHB + pl * 2048);
                b[pl][nt][1] = tr16(bbase + nt * CHB + pl * 2048 + 256);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                a[pl][mt][0] = tr16(abase + mt * CHB + pl * 2048);
                a[pl][mt][1] = tr16(abase + mt * CHB + pl * 2048 + 256);
            }
        }
    };
    auto multiply = [&](const v4s(&a)[NP][MT][2], const v4s(&b)[NP][NT][2]) {
        auto frag = [](const v4s(&f)[2]) { return __builtin_shufflevector(f[0], f[1], 0, 1, 2, 3, 4, 5, 6, 7); };
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int term = 0; term < NTERM; ++term)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    // smallest products first.  NP = 3: (0,2) (2,0) (1,1) (0,1) (1,0) (0,0); NP = 2: (0,1) (1,0) (0,0)
                    constexpr int PA3[6] = {0, 2, 1, 0, 1, 0}, PB3[6] = {2, 0, 1, 1, 0, 0}, PA2[3] = {0, 1, 0}, PB2[3] = {1, 0, 0};
                    const v8s av = frag(a[NP == 3 ? PA3[term] : NP == 2 ? PA2[term % 3] : 0][mt]), bv = frag(b[NP == 3 ? PB3[term] : NP == 2 ? PB2[term % 3] : 0][nt]);
                    if constexpr (NP != 2)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&av),
                                                                              *reinterpret_cast<const bf16x8 *>(&bv), acc[mt][nt], 0, 0, 0);
                    else
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8 *>(&av),
                                                                             *reinterpret_cast<const f16x8 *>(&bv), acc[mt][nt], 0, 0, 0);
                }
        __builtin_amdgcn_s_setprio(0);
    };

    // The main loop, once per loader role (the role is wave-uniform and fixed: three straight-line copies instead of one loop that
    // branches on it inside every slab).  One slab: wait for its pieces, barrier, read k-step 0 (pixels 0..15, pieces hf = 0) while --
    // from the second slab on -- k-step 1 of the PREVIOUS slab (already in a1 / b1) multiplies, then read k-step 1 (hf = 1, + 1024)
    // under the MFMAs of k-step 0.  The first slab is peeled off the loop: with an `if (s > 0)` inside it the compiler kept two
    // copies of the 64 accumulator registers and moved one into the other around every multiply (32 v_mov_b64 + MFMA-drain nops
    // per k-step in the ISA of round 3, and the loader's small arrays lived in scratch memory -- a scratch load with a vmcnt(0)
    // wait in the middle of every DMA issue: the loop ran 31 % MFMA-busy); a uniform body accumulates in place.
    auto run = [&](auto ROLE) __attribute__((always_inline)) {
        int issued = 0;
#pragma unroll
        for (int sg = 0; sg < NSTAGE - 1; ++sg)
            if (issued < total) {
                issue_slab(sg, ROLE);
                ++issued;
            }
        int cur = 0, nxt = NSTAGE - 1;
        if (total > 0) {
            if (NSTAGE >= 3 && issued - 1 >= 1) wwait_vmcnt<(NSTAGE >= 3 ? G : 0)>();
            else wwait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            load_frags(a_rd, b_rd, a0, b0);
            if (issued < total) {
                issue_slab(nxt, ROLE);
                ++issued;
            }
            __builtin_amdgcn_sched_barrier(0);
            load_frags(a_rd + 1024, b_rd + 1024, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            multiply(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            cur = cur + 1 == NSTAGE ? 0 : cur + 1;
            nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
        }
        for (int s = 1; s < total; ++s) {
            const int later = issued - s - 1;
            if (NSTAGE >= 3 && later >= 1) wwait_vmcnt<(NSTAGE >= 3 ? G : 0)>();
            else wwait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const unsigned so = (unsigned)(cur * STAGE);
            load_frags(a_rd + so, b_rd + so, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (issued < total) {
                issue_slab(nxt, ROLE);
                ++issued;
            }
            multiply(a1, b1);   // slab s-1 / k-step 1
            __builtin_amdgcn_sched_barrier(0);
            load_frags(a_rd + so + 1024, b_rd + so + 1024, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            multiply(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            cur = cur + 1 == NSTAGE ? 0 : cur + 1;
            nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
        }
    };
    if (loads_a) run(std::integral_constant<int, 0>{});
    else if (same_pix) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 2>{});
    if (total > 0) multiply(a1, b1);
    wwait_vmcnt<0>();
    if constexpr (NP == 2) {  // undo the operands' per-tensor power-of-two scales (exact)
        const float inv = *reinterpret_cast<const float *>(p.dy3 + p.dy_tr) * *reinterpret_cast<const float *>(p.x3 + p.x_tr);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] *= inv;
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

template <int BMK, int BNC, int WARPS_M, int WARPS_N, int NSTAGE, int NP> constexpr int wgrad_x3_smem() {
    return NSTAGE * ((BMK / 32) + (BNC / 32)) * (NP * 2048);
}

template <int BMK, int BNC, int WARPS_M, int WARPS_N, int NSTAGE, int NP = 3>
__global__ __launch_bounds__(64 * WARPS_M * WARPS_N, 2) void wgrad_x3_kernel(const WX3P p) {
    __shared__ __attribute__((aligned(16))) char smem[wgrad_x3_smem<BMK, BNC, WARPS_M, WARPS_N, NSTAGE, NP>()];
    wgrad_x3_body<BMK, BNC, WARPS_M, WARPS_N, NSTAGE, NP>(p, xcd_remap(blockIdx.x, gridDim.x), smem);
}

// GROUPED form: one launch computes the weight gradients of MANY conv layers (all deferred layers of a backward pass that
// share a tile class).  items[i] describes problem i, wg_begin[i] its first workgroup in the grid (wg_begin[n] = grid size);
// a workgroup finds its problem by binary search over the block-uniform table and runs the same body.  With hundreds of
// problems in one grid there are enough output tiles to fill the chip WITHOUT cutting the pixel reduction of the 8712-row
// layers into a dozen atomically-added splits: every workgroup runs the whole reduction of its tile (272 slabs instead of
// ~20), the per-launch fill / drain / atomic tails of 100 separate launches are paid once, and the problems are ordered
// longest-first so the long reductions of the early layers start at once.
template <int BMK, int BNC, int WARPS_M, int WARPS_N, int NSTAGE, int NP>
__global__ __launch_bounds__(64 * WARPS_M * WARPS_N, 2) void wgrad_x3_group_kernel(const WX3P *__restrict__ items, const int *__restrict__ wg_begin, int n) {
    __shared__ __attribute__((aligned(16))) char smem[wgrad_x3_smem<BMK, BNC, WARPS_M, WARPS_N, NSTAGE, NP>()];
    const int wgid = xcd_remap(blockIdx.x, gridDim.x);
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (wg_begin[mid] <= wgid) lo = mid; else hi = mid - 1;
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
    const WX3P p = items[lo];
    wgrad_x3_body<BMK, BNC, WARPS_M, WARPS_N, NSTAGE, NP>(p, wgid - wg_begin[lo], smem);
}

static long wx3_split(long base, long M, long target_wgs, long min_slabs) {
    if (dass_get_deterministic()) return 1;  // one pixel split: no cross-workgroup atomics (conv_igemm.hip)
    long want = (target_wgs + base - 1) / base;
    long maxsplit = M / (32 * min_slabs);
    if (maxsplit < 1) maxsplit = 1;
    if (want > maxsplit) want = maxsplit;
    return want < 1 ? 1 : want;
}

template <int BMK, int BNC, int WARPS_M, int WARPS_N, int NSTAGE, int NP = 3> int launch_wx3(WX3P &p, hipStream_t st, long target, long min_slabs) {
    p.ktiles = (p.K + BMK - 1) / BMK;
    p.ctiles = (p.C + BNC - 1) / BNC;
    const long base = (long)p.ktiles * p.ctiles * p.R * p.S;
    const long split = wx3_split(base, p.M, target, min_slabs);
    long pps = (p.M + split - 1) / split;
    pps = (pps + 31) / 32 * 32;
    p.pix_per_split = (int)pps;
    p.psplit = (int)((p.M + pps - 1) / pps);
    DASS_LAUNCH((wgrad_x3_kernel<BMK, BNC, WARPS_M, WARPS_N, NSTAGE, NP>), dim3((unsigned)(base * p.psplit)), dim3(64 * WARPS_M * WARPS_N), 0, st, p);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

}  // namespace

extern "C" int dass_conv2d_wgrad_x3(const void *x3, const void *dy3, float *dw, int N, int H, int W, int C, int OH, int OW, int K, int R,
                                    int S, int stride, int pad, int dil, int zero_first, void *stream) {
    if (!x3 || !dy3 || !dw) return DASS_ERR_ARG;
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0 || K <= 0 || R <= 0 || S <= 0 || stride < 1 || dil < 1) return DASS_ERR_ARG;
    if (((uintptr_t)x3 & 15) || ((uintptr_t)dy3 & 15)) return DASS_ERR_ARG;
    if ((long)N * OH * OW >= (1l << 31)) return DASS_ERR_ARG;
    const int CC = (C + 31) / 32, KC = (K + 31) / 32;
    const int parts = dass_get_x3_parts(), SB = parts * 64;
    const long xtr = x3_trailer_off((long)N * H * W, CC, parts), dtr = x3_trailer_off((long)N * OH * OW, KC, parts);
    const long xbytes = xtr + 16, dbytes = dtr + 16;
    if (xbytes >= (1l << 32) || dbytes >= (1l << 32)) return DASS_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (zero_first && hipMemsetAsync(dw, 0, sizeof(float) * (size_t)K * R * S * C, st) != hipSuccess) return DASS_ERR_LAUNCH;
    WX3P p;
    p.dy3 = (const char *)dy3;
    p.x3 = (const char *)x3;
    p.dw = dw;
    p.dy3_bytes = (unsigned)dbytes;
    p.x3_bytes = (unsigned)xbytes;
    p.dy_pitch = (unsigned)(KC * SB);
    p.x_pitch = (unsigned)(CC * SB);
    p.dy_zero = (unsigned)((long)N * OH * OW * KC * SB);
    p.x_zero = (unsigned)((long)N * H * W * CC * SB);
    p.dy_tr = (unsigned)dtr;
    p.x_tr = (unsigned)xtr;
    p.N = N; p.H = H; p.W = W; p.C = C; p.OH = OH; p.OW = OW; p.K = K; p.R = R; p.S = S;
    p.stride = stride; p.pad = pad; p.dil = dil;
    p.M = N * OH * OW;
    p.KC = KC; p.CC = CC;
    x3_set_magic(OH * OW, p.mg_ohw, p.sh_ohw);
    x3_set_magic(OW, p.mg_ow, p.sh_ow);
    static const long target = getenv("DASS_WX3_TARGET") ? atol(getenv("DASS_WX3_TARGET")) : 768;
    static const long min_slabs = getenv("DASS_WX3_MINSLABS") ? atol(getenv("DASS_WX3_MINSLABS")) : 16;
    static const int force = getenv("DASS_WX3_TILE") ? atoi(getenv("DASS_WX3_TILE")) : 0;
    const bool big = (K > 64 && C > 64 && force != 2) || force == 1;
    if (parts == 2) {
        if (big) return launch_wx3<128, 128, 4, 2, 3, 2>(p, st, target, min_slabs);
        return launch_wx3<64, 64, 2, 2, 3, 2>(p, st, target, min_slabs);
    }
    if (parts == 1) {
        if (big) return launch_wx3<128, 128, 2, 2, 2, 1>(p, st, target, min_slabs);
        return launch_wx3<64, 64, 2, 2, 3, 1>(p, st, target, min_slabs);
    }
    if (big) return launch_wx3<128, 128, 4, 2, 3>(p, st, target, min_slabs);
    return launch_wx3<64, 64, 2, 2, 3>(p, st, target, min_slabs);
}

// ---- grouped launch (see wgrad_x3_group_kernel).  items: n x 16 int64 on the HOST
//   {x3, dy3, dw, N, H, W, C, OH, OW, K, R, S, stride, pad, dil, 0};  dw must be zeroed (tiles are ADDED: big-M problems are still cut
// into pixel ranges of at most DASS_WX3_GROUP_SLABS 32-pixel slabs, default 96 (512 while one launch ran alone at the end of
// backward; the chunked launches that now share the chip with the rest of backward want short workgroups: 31.4 -> 30.0 ms per step), so no workgroup runs much longer than the
// rest).  scratch: device buffer of dass_conv2d_wgrad_x3_group_scratch_bytes(n) for the problem table (written by an
// asynchronous copy on `stream`: keep it alive until the launches have run, do not reuse it for another call before that).
extern "C" int64_t dass_conv2d_wgrad_x3_group_scratch_bytes(int n) { return (int64_t)(n > 0 ? n : 1) * (sizeof(WX3P) + 8) + 64; }

// ---- staging tables of CAPTURED grouped launches (a hipGraph around the train step, dass_hip/graph.py).  The copy of the problem table
// becomes a graph node that reads its pinned source at EVERY replay, so a table used inside a capture must stay untouched for as long
// as the graph lives -- and not longer: the tables of a capture belong to a token, opened BEFORE the capture starts
// (dass_graph_capture_open reserves the pinned buffers there: nothing may be allocated while a stream captures) and handed back when
// the graph is destroyed or its capture failed (dass_graph_release).  One pool per process, shared by every thread and every tile
// class (autograd runs backward on its own threads).
namespace {
struct CapSlot {
    char *p = nullptr;   // WX3P[cap]
    int *b = nullptr;    // int[cap + 1]
    int cap = 0;
    int64_t token = 0;   // 0 = free
};
constexpr int CAP_SLOT_ITEMS = 1024;
std::mutex g_cap_mu;
std::vector<CapSlot> g_cap_slots;
int64_t g_cap_token = 0, g_cap_open = 0;
}  // namespace

extern "C" int64_t dass_graph_capture_open(int slots) {
    std::lock_guard<std::mutex> lk(g_cap_mu);
    int free_slots = 0;
    for (const CapSlot &c : g_cap_slots) free_slots += c.token == 0;
    for (; free_slots < slots; ++free_slots) {
        CapSlot c;
        c.cap = CAP_SLOT_ITEMS;
        if (hipHostMalloc((void **)&c.p, sizeof(WX3P) * c.cap) != hipSuccess) return -1;
        if (hipHostMalloc((void **)&c.b, sizeof(int) * (c.cap + 1)) != hipSuccess) { (void)hipHostFree(c.p); return -1; }
        g_cap_slots.push_back(c);
    }
    g_cap_open = ++g_cap_token;
    return g_cap_open;
}

extern "C" int dass_graph_capture_close(void) {
    std::lock_guard<std::mutex> lk(g_cap_mu);
    g_cap_open = 0;
    return DASS_OK;
}

extern "C" int dass_graph_release(int64_t token) {
    if (token <= 0) return DASS_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_cap_mu);
    int n = 0;
    for (CapSlot &c : g_cap_slots)
        if (c.token == token) { c.token = 0; ++n; }
    return n;
}

/* staging tables currently owned by captured graphs / free (diagnostics, tests) */
extern "C" int dass_graph_slots(int *owned, int *free_slots) {
    std::lock_guard<std::mutex> lk(g_cap_mu);
    int o = 0, f = 0;
    for (const CapSlot &c : g_cap_slots) (c.token ? o : f) += 1;
    if (owned) *owned = o;
    if (free_slots) *free_slots = f;
    return DASS_OK;
}

namespace {
struct GroupItem {
    WX3P p;
    long wgs;
    int slabs;
};

template <int BMK, int BNC, int WARPS_M, int WARPS_N, int NSTAGE, int NP>
int launch_group(GroupItem *it, int n, char *scratch, hipStream_t st) {
    if (n == 0) return DASS_OK;
    // longest reductions first: the hardware hands out workgroups in grid order
    for (int i = 1; i < n; ++i) {  // insertion sort (n is a few hundred)
        GroupItem key = it[i];
        int j = i - 1;
        while (j >= 0 && it[j].slabs < key.slabs) {
            it[j + 1] = it[j];
            --j;
        }
        it[j + 1] = key;
    }
    // pinned staging tables (truly asynchronous copies), a ring of 64 per thread and tile class: a slot is rewritten only after the copy
    // out of it, issued 64 grouped launches earlier, has completed -- the host never waits for the stream in steady state.
    // Under stream capture the table comes from the capture's own pool instead (dass_graph_capture_open above).
    struct Slot {
        WX3P *p = nullptr;
        int *b = nullptr;
        int cap = 0;
        hipEvent_t done = nullptr;
        bool used = false;
    };
    constexpr int NSLOT = 64;
    static thread_local Slot ring[NSLOT];
    static thread_local int next_slot = 0;
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cap_status) == hipSuccess && cap_status == hipStreamCaptureStatusActive;
    Slot cap_sl;
    Slot *slp = nullptr;
    if (capturing) {
        std::lock_guard<std::mutex> lk(g_cap_mu);
        if (g_cap_open == 0) {
            fprintf(stderr, "libdass_hip: a grouped weight-gradient launch is being captured into a hipGraph without dass_graph_capture_open() "
                            "(dass_hip.graph.GraphedStep calls it): no staging table may be allocated during a capture\n");
            return DASS_ERR_UNSUPPORTED;
        }
        for (CapSlot &c : g_cap_slots)
            if (c.token == 0 && c.cap >= n) {
                c.token = g_cap_open;
                cap_sl.p = reinterpret_cast<WX3P *>(c.p);
                cap_sl.b = c.b;
                cap_sl.cap = c.cap;
                slp = &cap_sl;
                break;
            }
        if (!slp) {
            fprintf(stderr, "libdass_hip: the hipGraph capture needs more staging tables than dass_graph_capture_open() reserved (%d exist, all "
                            "owned by live graphs; release finished graphs or reserve more)\n", (int)g_cap_slots.size());
            return DASS_ERR_UNSUPPORTED;
        }
    } else {
        slp = &ring[next_slot];
        next_slot = (next_slot + 1) % NSLOT;
        if (slp->used && hipEventSynchronize(slp->done) != hipSuccess) return DASS_ERR_LAUNCH;
        if (slp->cap < n) {
            if (slp->p) { (void)hipHostFree(slp->p); (void)hipHostFree(slp->b); }
            slp->cap = n > 448 ? n + 64 : 512;
            if (hipHostMalloc((void **)&slp->p, sizeof(WX3P) * slp->cap) != hipSuccess || hipHostMalloc((void **)&slp->b, sizeof(int) * (slp->cap + 1)) != hipSuccess) {
                slp->cap = 0;
                slp->p = nullptr;
                return DASS_ERR_LAUNCH;
            }
        }
        if (!slp->done && hipEventCreateWithFlags(&slp->done, hipEventDisableTiming) != hipSuccess) return DASS_ERR_LAUNCH;
    }
    Slot &sl = *slp;
    WX3P *host_p = sl.p;
    int *host_b = sl.b;
    long total = 0;
    for (int i = 0; i < n; ++i) {
        host_p[i] = it[i].p;
        host_b[i] = (int)total;
        total += it[i].wgs;
    }
    host_b[n] = (int)total;
    if (total >= (1l << 31)) return DASS_ERR_UNSUPPORTED;
    WX3P *dev_p = reinterpret_cast<WX3P *>(scratch);
    int *dev_b = reinterpret_cast<int *>(scratch + sizeof(WX3P) * n);
    if (hipMemcpyAsync(dev_p, host_p, sizeof(WX3P) * n, hipMemcpyHostToDevice, st) != hipSuccess) return DASS_ERR_LAUNCH;
    if (hipMemcpyAsync(dev_b, host_b, sizeof(int) * (n + 1), hipMemcpyHostToDevice, st) != hipSuccess) return DASS_ERR_LAUNCH;
    if (!capturing) {
        if (hipEventRecord(sl.done, st) != hipSuccess) return DASS_ERR_LAUNCH;
        sl.used = true;
    }
    // (DASS_WX3_GROUP_LDS_PAD: bytes of dynamic LDS added to the grouped launch.  The weight gradients share the chip with backward's chain of input-
    //  gradient launches: two resident workgroups of 64 KB leave no room for a conv workgroup (33 KB) on that CU; padded beyond 80 KB only ONE fits and two
    //  conv workgroups keep running beside it)
    static const int lds_pad = getenv("DASS_WX3_GROUP_LDS_PAD") ? atoi(getenv("DASS_WX3_GROUP_LDS_PAD")) : 0;
    DASS_LAUNCH((wgrad_x3_group_kernel<BMK, BNC, WARPS_M, WARPS_N, NSTAGE, NP>), dim3((unsigned)total), dim3(64 * WARPS_M * WARPS_N), lds_pad, st, dev_p, dev_b, n);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}
}  // namespace

extern "C" int dass_conv2d_wgrad_x3_group(const int64_t *items, int n, void *scratch, int64_t scratch_bytes, void *stream) {
    if (!items || n <= 0 || !scratch || ((uintptr_t)scratch & 15)) return DASS_ERR_ARG;
    if (scratch_bytes < dass_conv2d_wgrad_x3_group_scratch_bytes(n)) return DASS_ERR_ARG;
    const int parts = dass_get_x3_parts(), SB = parts * 64;
    static const long cap_slabs = getenv("DASS_WX3_GROUP_SLABS") ? atol(getenv("DASS_WX3_GROUP_SLABS")) : 96;
    GroupItem *big = new GroupItem[n], *small = new GroupItem[n];
    int nb = 0, ns = 0, rc = DASS_OK;
    for (int i = 0; i < n && rc == DASS_OK; ++i) {
        const int64_t *d = items + 16 * i;
        const void *x3 = (const void *)d[0], *dy3 = (const void *)d[1];
        float *dw = (float *)d[2];
        const int N = (int)d[3], H = (int)d[4], W = (int)d[5], C = (int)d[6], OH = (int)d[7], OW = (int)d[8], K = (int)d[9], R = (int)d[10],
                  S = (int)d[11], stride = (int)d[12], pad = (int)d[13], dil = (int)d[14];
        if (!x3 || !dy3 || !dw || ((uintptr_t)x3 & 15) || ((uintptr_t)dy3 & 15)) rc = DASS_ERR_ARG;
        if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0 || K <= 0 || R <= 0 || S <= 0 || stride < 1 || dil < 1) rc = DASS_ERR_ARG;
        if (rc != DASS_OK) break;
        if ((long)N * OH * OW >= (1l << 31)) { rc = DASS_ERR_ARG; break; }
        const int CC = (C + 31) / 32, KC = (K + 31) / 32;
        const long xtr = x3_trailer_off((long)N * H * W, CC, parts), dtr = x3_trailer_off((long)N * OH * OW, KC, parts);
        if (xtr + 16 >= (1l << 32) || dtr + 16 >= (1l << 32)) { rc = DASS_ERR_UNSUPPORTED; break; }
        const bool is_big = K > 64 && C > 64;
        GroupItem &g = is_big ? big[nb++] : small[ns++];
        WX3P &p = g.p;
        p.dy3 = (const char *)dy3; p.x3 = (const char *)x3; p.dw = dw;
        p.dy3_bytes = (unsigned)(dtr + 16); p.x3_bytes = (unsigned)(xtr + 16);
        p.dy_pitch = (unsigned)(KC * SB); p.x_pitch = (unsigned)(CC * SB);
        p.dy_zero = (unsigned)((long)N * OH * OW * KC * SB); p.x_zero = (unsigned)((long)N * H * W * CC * SB);
        p.dy_tr = (unsigned)dtr; p.x_tr = (unsigned)xtr;
        p.N = N; p.H = H; p.W = W; p.C = C; p.OH = OH; p.OW = OW; p.K = K; p.R = R; p.S = S;
        p.stride = stride; p.pad = pad; p.dil = dil;
        p.M = N * OH * OW; p.KC = KC; p.CC = CC;
        x3_set_magic(OH * OW, p.mg_ohw, p.sh_ohw);
        x3_set_magic(OW, p.mg_ow, p.sh_ow);
        const int tile = is_big ? 128 : 64;
        p.ktiles = (K + tile - 1) / tile;
        p.ctiles = (C + tile - 1) / tile;
        long split = dass_get_deterministic() ? 1 : ((long)p.M + 32 * cap_slabs - 1) / (32 * cap_slabs);
        long pps = (p.M + split - 1) / split;
        pps = (pps + 31) / 32 * 32;
        p.pix_per_split = (int)pps;
        p.psplit = (int)((p.M + pps - 1) / pps);
        g.wgs = (long)p.ktiles * p.ctiles * R * S * p.psplit;
        g.slabs = (int)(pps / 32);
    }
    hipStream_t st = (hipStream_t)stream;
    char *sc = (char *)scratch;
    static const int gtile = getenv("DASS_WX3_GROUP_TILE") ? atoi(getenv("DASS_WX3_GROUP_TILE")) : 1;  // tuning knob (measured: 6.9 / 10.7 / 9.3 / 10.1 ms per R101 step for 1 / 2 / 3 / 0)
    if (rc == DASS_OK) {
        if (parts == 2) {
            if (gtile == 1) rc = launch_group<128, 128, 2, 2, 2, 2>(big, nb, sc, st);       // 4 waves of 64 x 64, two 32 KB stages: two workgroups per CU
            else if (gtile == 2) rc = launch_group<128, 128, 2, 2, 3, 2>(big, nb, sc, st);  // the same with three stages (one workgroup per CU)
            else if (gtile == 3) rc = launch_group<128, 128, 4, 2, 2, 2>(big, nb, sc, st);  // 8 waves of 32 x 64, two stages
            else rc = launch_group<128, 128, 4, 2, 3, 2>(big, nb, sc, st);
        } else if (parts == 1) {
            rc = launch_group<128, 128, 2, 2, 2, 1>(big, nb, sc, st);
        } else {
            rc = launch_group<128, 128, 4, 2, 3, 3>(big, nb, sc, st);
        }
    }
    if (rc == DASS_OK) {
        char *sc2 = sc + ((sizeof(WX3P) * nb + sizeof(int) * (nb + 1) + 63) / 64) * 64;
        if (parts == 2) rc = launch_group<64, 64, 2, 2, 3, 2>(small, ns, sc2, st);
        else if (parts == 1) rc = launch_group<64, 64, 2, 2, 3, 1>(small, ns, sc2, st);
        else rc = launch_group<64, 64, 2, 2, 3, 3>(small, ns, sc2, st);
    }
    delete[] big;
    delete[] small;
    return rc;
}
