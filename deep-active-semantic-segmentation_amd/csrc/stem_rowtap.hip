// Network stems on exact f32 MFMAs (reference: models/backbone/resnet.py:65 7x7/s2 over the 3-channel image, mobilenet.py:14 and
// xception.py stems 3x3/s2): the forward conv and its weight gradient behind dass_conv2d_rowtap / dass_conv2d_rowtap_wgrad.
//
// The image is a DENSE NHWC tensor [N][H][W][Cin]; one kernel row ("tap row") of a pixel's window is S * Cin contiguous floats, the
// whole window R such runs, L = R * S * Cin values (147 for the 7x7 stem).  Both kernels are bounded by the f32 MFMA pipe
// (v_mfma_f32_32x32x2_f32, 256 FLOP / clk / CU) and by one pass over the [M][K] activation, so neither stages anything through
// LDS for the streaming operand:
//   * forward: a wave owns 32 output pixels x all K channels; the A fragment (pixel = lane & 31, window index = 2 step + (lane >> 5)) is
//     one predicated dword load per lane and step (the 147-value window is re-read from L1 / L2: the image is 25 MB), the B fragment
//     comes from the weight table in LDS.  The 9.9 GFLOP of the R101 stem cost 229 us on the generic implicit-GEMM kernel (window
//     padded 21 -> 32 per tap row, bf16 splits in the loop); here they are 74 steps x 2 MFMAs per 32 pixels.
//   * weight gradient: dW[k][c] = sum over pixels dy[pixel][k] * window[pixel][c].  The generic kernel ran one workgroup per tap row
//     (each re-reading dy: 7 x 135 MB) on 64-wide column tiles holding 21 columns; here a wave keeps the WHOLE K x L gradient
//     (2 x 5 accumulator blocks of 32 x 32) in registers and streams its share of the pixels once: dy is read once, the columns
//     are packed (147 of 160 used), and the four waves of a workgroup fold through LDS before one set of atomics.
#include <hip/hip_runtime.h>

#include "dass_common.h"
#include "../../include/dass_hip.h"

namespace {

using f32x16 = __attribute__((__vector_size__(16 * sizeof(float)))) float;

struct RowtapP {
    const float *x, *w, *dy;
    float *y, *dw;
    long ldy;
    int N, H, W, Cin, OH, OW, K, R, SC, L, stride, pad, M;
    unsigned mg_ohw, mg_ow;
    int sh_ohw, sh_ow;
    int nblk;   // forward: 32-pixel blocks
    int chunk;  // weight gradient: pixels per wave (even)
    int adv_px, adv_row, adv_img;  // weight gradient: byte steps of a window's first element for +2 pixels / a wrapped row / a wrapped image
};

__device__ __forceinline__ unsigned umin(unsigned a, unsigned b) { return a < b ? a : b; }

constexpr int PAD_ROW = 1 << 14;  // window row of the table's padding entries: fails every `row < H` test (host checks H < PAD_ROW)

constexpr int FWD_WAVES = 8;  // waves per workgroup (they share one weight table)
constexpr int WG_WAVES = 8;   // weight gradient: waves per workgroup (they fold through LDS before the atomics)

struct __attribute__((packed, aligned(4))) f3 { float v[3]; };  // one pixel of the 3-channel image: 12 B, dword-aligned

// Forward, 3-channel images.  A wave owns 32 output pixels (lane & 31).  A tap row of a pixel's window is S pixels = 3 S contiguous
// floats; the two half-waves of the MFMA's k index take PX = ceil(S / 2) pixels of it each (half 1 starts at pixel S - PX; where the
// two ranges overlap, half 1's weights are zero), so a lane reads its share of a tap row as PX 12-byte loads instead of 3 PX dword
// loads: a third of the L1 accesses for the same lines (a wave's loads stride 24 B between lanes, ~12 lines per instruction either
// way; with dword loads the kernel ran at the L1's line rate, 164 us at 8 x 513^2, not at the MFMA's).  The next tap row's loads fly
// under this row's 3 PX x NT MFMAs.
template <int NT, int PX> __global__ __launch_bounds__(64 * FWD_WAVES) void rowtap_fwd_kernel(const RowtapP p) {
    constexpr int NTH = 64 * FWD_WAVES, SPR = 3 * PX, LDW = NT * 32;  // steps per tap row; floats per table row
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *wl = smem;  // [R][SPR][2 halves][LDW], columns XOR-ed with 32 * half (the two half-waves of a B read hit disjoint banks)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int S = p.SC / 3, h1 = S - PX;  // half 1's first pixel
    for (int i = tid; i < p.R * SPR * 2 * LDW; i += NTH) {
        const int k = i % LDW, hs = i / LDW, h = hs & 1, st = hs >> 1;
        const int r = st / SPR, s = st - r * SPR, px = (h ? h1 : 0) + s / 3, ch = s - (s / 3) * 3;
        const bool dup = h && (s / 3) < 2 * PX - S;  // pixel already covered by half 0
        wl[hs * LDW + (k ^ (NT == 2 ? h * 32 : 0))] = dup ? 0.f : p.w[(long)k * p.L + r * p.SC + px * 3 + ch];
    }
    __syncthreads();
    const int ohw = p.OH * p.OW;
    for (int blk = blockIdx.x * FWD_WAVES + wave; blk < p.nblk; blk += gridDim.x * FWD_WAVES) {
        const int pix = blk * 32 + l31;
        const bool pok = pix < p.M;
        const int pp = pok ? pix : 0;
        const int n = x3_fastdiv(pp, p.mg_ohw, p.sh_ohw);
        const int rem = pp - n * ohw;
        const int oh = x3_fastdiv(rem, p.mg_ow, p.sh_ow);
        const int ow = rem - oh * p.OW;
        const int iy0 = pok ? oh * p.stride - p.pad : -PAD_ROW * 2, ix0 = ow * p.stride - p.pad + (half ? h1 : 0);
        const float *xb = p.x + ((long)(n * p.H + iy0) * p.W + ix0) * 3;  // (dereferenced only where the window is inside the image)
        bool colok[PX];
#pragma unroll
        for (int j = 0; j < PX; ++j) colok[j] = (unsigned)(ix0 + j) < (unsigned)p.W;
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
        auto fetch = [&](int r, f3 (&a)[PX]) {
            const bool rowok = (unsigned)(iy0 + r) < (unsigned)p.H;
            const float *row = xb + (long)r * p.W * 3;
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                // (unconditional load from a clamped address + select: a predicated load becomes a branch with its own
                // s_waitcnt, which serialises the loads against the MFMAs)
                const bool ok = rowok & colok[j];
                const f3 v = *reinterpret_cast<const f3 *>(ok ? row + j * 3 : p.x);
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) a[j].v[ch] = ok ? v.v[ch] : 0.f;
            }
        };
        f3 a_cur[PX], a_nxt[PX];
        fetch(0, a_cur);
        for (int r = 0; r < p.R; ++r) {
            if (r + 1 < p.R) fetch(r + 1, a_nxt);
            const float *wr = wl + (r * SPR * 2 + half) * LDW + l31;
            const int col_of[2] = {NT == 2 ? half * 32 : 0, NT == 2 ? (half ^ 1) * 32 : 0};  // (the table's XOR swizzle)
#pragma unroll
            for (int j = 0; j < PX; ++j)
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const int s = j * 3 + ch;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[j].v[ch], wr[s * 2 * LDW + col_of[nt]], acc[nt], 0, 0, 0);
                }
#pragma unroll
            for (int j = 0; j < PX; ++j) a_cur[j] = a_nxt[j];
        }
        // D layout: column (channel) = lane & 31, row (pixel) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): a half-wave stores 128 B runs
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = blk * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                if (row < p.M) p.y[(long)row * p.ldy + nt * 32 + l31] = acc[nt][reg];
            }
    }
}

template <int MT, int JB> __global__ __launch_bounds__(64 * WG_WAVES, 1) void rowtap_wgrad_kernel(const RowtapP p) {
    constexpr int U = 2;  // pixel pairs per round of loads
    constexpr int LDR = JB * 32, NTH = 64 * WG_WAVES;
    __shared__ float red[MT * 32 * LDR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    // per lane: its JB window columns (tap row, pixel inside the row, byte offset from the window's first element)
    int off[JB], wrow[JB], wcol[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        const int c = j * 32 + l31;
        const int r = c / p.SC, q = c - r * p.SC;
        off[j] = c < p.L ? (r * p.W * p.Cin + q) * 4 : 0;
        wrow[j] = c < p.L ? r : PAD_ROW;
        wcol[j] = q / p.Cin;
    }
    f32x16 acc[MT][JB];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const long gw = (long)blockIdx.x * WG_WAVES + wave;
    const long pbeg = gw * p.chunk;
    long pend = pbeg + p.chunk;
    if (pend > p.M) pend = p.M;
    // this half-wave's pixel cursor (pixels pbeg + half, + 2, + 4, ...): carried with adds only -- the divisions and 64-bit
    // multiplies of a per-round re-derivation are quarter-rate VALU work that, at ~150 instructions per 20 MFMAs, kept the
    // MFMA pipe 40 % busy.  All offsets are bytes in 32 bits (host: image and dy below 2^31 bytes... checked there).
    int left, ow, oh, iy0, ix0, xoff, dyoff;
    {
        const long pix = pbeg + half;
        left = (int)(pend - pix);  // > 0: the cursor's pixel exists
        const int pp = left > 0 ? (int)pix : 0;
        const int n = pp / (p.OH * p.OW), rem = pp - n * (p.OH * p.OW);
        oh = rem / p.OW;
        ow = rem - oh * p.OW;
        iy0 = oh * p.stride - p.pad;
        ix0 = ow * p.stride - p.pad;
        xoff = (((n * p.H + iy0) * p.W) + ix0) * p.Cin * 4;
        dyoff = pp * (int)p.ldy * 4 + l31 * 4;
    }
    const char *xbase = reinterpret_cast<const char *>(p.x), *dybase = reinterpret_cast<const char *>(p.dy);
    const int dy_step = 2 * (int)p.ldy * 4;
    const unsigned dy_last = (unsigned)(p.M - 1) * (unsigned)p.ldy * 4u + l31 * 4;
    auto fetch = [&](float (&a)[U][MT], float (&b)[U][JB]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool pok = left > 0;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const float v = *reinterpret_cast<const float *>(dybase + umin((unsigned)dyoff, dy_last) + i * 128);  // (clamped: always a valid row)
                a[u][i] = pok ? v : 0.f;
            }
#pragma unroll
            for (int j = 0; j < JB; ++j) {
                // (unconditional load from a clamped address + select: see the forward kernel)
                const bool ok = pok & ((unsigned)(iy0 + wrow[j]) < (unsigned)p.H) & ((unsigned)(ix0 + wcol[j]) < (unsigned)p.W);  // (&: no short-circuit branches)
                const float v = *reinterpret_cast<const float *>(xbase + (unsigned)(ok ? xoff + off[j] : 0));
                b[u][j] = ok ? v : 0.f;
            }
            // + 2 pixels
            left -= 2;
            dyoff += dy_step;
            ow += 2;
            ix0 += 2 * p.stride;
            xoff += p.adv_px;
            // (selects, not branches: the two half-waves wrap at different times, and a divergent branch here splits the round)
            const bool wr = ow >= p.OW;
            ow -= wr ? p.OW : 0;
            ix0 -= wr ? p.OW * p.stride : 0;
            xoff += wr ? p.adv_row : 0;
            oh += wr ? 1 : 0;
            iy0 += wr ? p.stride : 0;
            const bool wi = oh >= p.OH;
            oh = wi ? 0 : oh;
            iy0 -= wi ? p.OH * p.stride : 0;
            xoff += wi ? p.adv_img : 0;
        }
    };
    auto mma = [&](const float (&a)[U][MT], const float (&b)[U][JB]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < JB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
    };
    // three rounds in registers: the loads of round i + 2 are issued before the MFMAs of round i (dy comes from HBM: one round of
    // MFMAs, 1280 cycles, does not cover that latency; two do).  Rounds past the end load nothing and multiply zeros.
    float a0[U][MT], b0[U][JB], a1[U][MT], b1[U][JB], a2[U][MT], b2[U][JB];
    constexpr int RP = 2 * U;  // pixels per round
    fetch(a0, b0);
    fetch(a1, b1);
    for (long p0 = pbeg; p0 < pend; p0 += 3 * RP) {
        fetch(a2, b2);
        mma(a0, b0);
        fetch(a0, b0);
        mma(a1, b1);
        fetch(a1, b1);
        mma(a2, b2);
    }
    // the waves hold partial gradients over disjoint pixels: fold them in wave order (a fixed order), then one atomic per value
    for (int wv = 0; wv < WG_WAVES; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < JB; ++j)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        float *slot = red + (i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half) * LDR + j * 32 + l31;
                        *slot = wv == 0 ? acc[i][j][reg] : *slot + acc[i][j][reg];
                    }
        }
        __syncthreads();
    }
    for (int i = tid; i < p.K * p.L; i += NTH) {
        const int k = i / p.L, c = i - k * p.L;
        atomicAdd(p.dw + i, red[k * LDR + c]);
    }
}

int cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return n;
}

bool fast_enabled() {
    const char *e = getenv("DASS_ROWTAP_FAST");  // (0: the generic implicit-GEMM kernels, for A/B timing)
    return !(e && e[0] == '0');
}

bool fill(RowtapP &p, int N, int H, int W, int Cin, int OH, int OW, int K, int R, int S, int stride, int pad) {
    if (H >= PAD_ROW || W >= (1 << 15) || (long)N * H * W * Cin >= (1l << 31) || (long)N * OH * OW >= (1l << 31) - 64) return false;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.K = K; p.R = R; p.SC = S * Cin; p.L = R * S * Cin;
    p.stride = stride; p.pad = pad; p.M = N * OH * OW;
    x3_set_magic(OH * OW, p.mg_ohw, p.sh_ohw);
    x3_set_magic(OW, p.mg_ow, p.sh_ow);
    return true;
}

}  // namespace

// 1: launched; 0: shape outside the specialisation (the caller runs the generic kernel); < 0: launch error
int dass_rowtap_fwd_fast(const float *x, const float *w, float *y, long ldy, int N, int H, int W, int Cin, int OH, int OW, int K, int R, int S,
                         int stride, int pad, hipStream_t st) {
    if (!fast_enabled() || (K != 32 && K != 64) || Cin != 3 || (S != 7 && S != 3) || R > 7) return 0;
    RowtapP p{};
    if (!fill(p, N, H, W, Cin, OH, OW, K, R, S, stride, pad)) return 0;
    p.x = x; p.w = w; p.y = y; p.ldy = ldy;
    p.nblk = (p.M + 31) / 32;
    const int px = (S + 1) / 2;
    const size_t lds = (size_t)R * 3 * px * 2 * K * 4;  // 7 x 12 x 2 x 64 floats = 43 KB
    int grid = (p.nblk + FWD_WAVES - 1) / FWD_WAVES;
    if (grid > 2 * cus()) grid = 2 * cus();
    const dim3 blk(64 * FWD_WAVES);
    if (K == 64 && S == 7) DASS_LAUNCH((rowtap_fwd_kernel<2, 4>), dim3(grid), blk, lds, st, p);
    else if (K == 64) DASS_LAUNCH((rowtap_fwd_kernel<2, 2>), dim3(grid), blk, lds, st, p);
    else if (S == 7) DASS_LAUNCH((rowtap_fwd_kernel<1, 4>), dim3(grid), blk, lds, st, p);
    else DASS_LAUNCH((rowtap_fwd_kernel<1, 2>), dim3(grid), blk, lds, st, p);
    return hipGetLastError() == hipSuccess ? 1 : -1;
}

int dass_rowtap_wgrad_fast(const float *x, const float *dy, long lddy, float *dw, int N, int H, int W, int Cin, int OH, int OW, int K, int R, int S,
                           int stride, int pad, hipStream_t st) {
    const int L = R * S * Cin;
    // (dass_set_deterministic(1): the generic kernel with ONE pixel split -- this kernel's workgroups meet in atomics)
    if (!fast_enabled() || dass_get_deterministic() || !((K == 64 && L <= 160) || (K == 32 && L <= 32))) return 0;
    RowtapP p{};
    if (!fill(p, N, H, W, Cin, OH, OW, K, R, S, stride, pad)) return 0;
    p.x = x; p.dy = dy; p.dw = dw; p.ldy = lddy;
    const int waves = cus() * WG_WAVES;
    long chunk = ((long)p.M + waves - 1) / waves;
    chunk = (chunk + 3) / 4 * 4;  // (whole rounds of 4 pixels: only a wave's last round is ragged)
    p.chunk = (int)chunk;
    if (OW < 2 || (long)p.M * lddy * 4 >= (1l << 31) || (long)N * H * W * Cin * 4 >= (1l << 31)) return 0;  // (32-bit byte offsets, +2-pixel cursor)
    p.adv_px = 2 * stride * Cin * 4;
    p.adv_row = (stride * W - OW * stride) * Cin * 4;
    p.adv_img = (H - OH * stride) * W * Cin * 4;
    const int grid = (int)(((long)p.M + chunk * WG_WAVES - 1) / (chunk * WG_WAVES));
    if (K == 64) DASS_LAUNCH((rowtap_wgrad_kernel<2, 5>), dim3(grid), dim3(64 * WG_WAVES), 0, st, p);
    else DASS_LAUNCH((rowtap_wgrad_kernel<1, 1>), dim3(grid), dim3(64 * WG_WAVES), 0, st, p);
    return hipGetLastError() == hipSuccess ? 1 : -1;
}
