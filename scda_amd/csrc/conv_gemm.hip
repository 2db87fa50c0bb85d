// conv_gemm.hip -- fp32-MFMA implicit-GEMM convolution (forward, data-gradient,
// weight-gradient) and dense GEMM for gfx950, behind the C ABI of scda_ops.h.
//
// GEMM view (NCHW tensors, no layout change at the boundary):
//   forward   Y[co][n,oy,ox]   = sum_{ci,kh,kw} W[co][ci,kh,kw] * X[n,ci,oy*S+kh-P,ox*S+kw-P]
//             M = Cout, N = batch*OH*OW (pixels, contiguous in memory), K = Cin*KH*KW
//   dgrad     dX[ci][n,iy,ix]  = sum_{co,kh,kw} Wt[ci][co,kh,kw] * dY[n,co,(iy+P-kh)/S,(ix+P-kw)/S]
//             same kernel, gather predicate differs (DGRAD); Wt = W with dims 0/1 swapped
//   wgrad     dW[co][ci,kh,kw] = sum_{n,oy,ox} dY[n,co,oy,ox] * X[n,ci,oy*S+kh-P,ox*S+kw-P]
//             M = Cout, N = Cin*KH*KW, K = batch*OH*OW, always split-K (deterministic
//             partial slabs + fixed-order reduce, no atomics)
// The im2col matrix is never materialised: each thread gathers its B elements
// straight from the activation tensor into registers (prefetch for the next
// K-step), then stages them in LDS in the k-major layout mfma_tile.h reads.
//
// K order for forward / dgrad.  A 16-deep K-slab is 16 consecutive channels at ONE filter tap, so the (kh,kw) decode,
// the bounds test and the pixel offset are computed once per slab (scalar / one VALU op each) and the 8 gathers of a
// thread are `base + j*2*plane` -- the ablation in scripts/ablate/ showed the per-element index math of a plain
// channel-major order cost 35 % of the kernel.  Slabs are ordered CHANNEL-BLOCK major, tap minor:
//     k = ((c/16) * KH*KW + tap) * 16 + c%16          (channel count % 16 == 0; otherwise tap-major k = tap*C + c)
// so the 9 taps of one 16-channel group are consecutive K-steps and re-read the same input lines within a few hundred
// cycles (L1/L2 hits).  With tap-major slabs the reuse distance was C/16 slabs x all resident workgroups = 3..25 MB per
// XCD, more than its 4 MB L2: rocprofv3 FETCH_SIZE showed conv1_2 fetching 7x its input (profiles/r01_pmc_traffic.md).
// The weight operand is pre-permuted to match by scda_conv2d_pack_weight_hip (cached per optimiser step by the caller).
//
// Workgroup -> tile mapping (all three kernels): 1-D grid, XCD-aware.  The dispatcher places workgroup b on XCD b % 8;
// `tile_coords` gives each XCD a CONTIGUOUS run of logical tile ids (bijective for any grid size) ordered M-tile
// fastest, then N-tile, then K-split, so the workgroups that share an input tile (all M-tiles of one pixel tile, and
// vertically adjacent pixel tiles) run on the same XCD at about the same time and share its L2.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "mfma_tile.h"
#include "wino_pack.h"

namespace scda {

static bool xcd_swizzle_enabled() {
    static const bool on = [] { const char *v = getenv("SCDA_XCD_SWIZZLE"); return !(v && atoi(v) == 0); }();
    return on;
}

// logical tile of this workgroup: (tx = N-tile, ty = M-tile, tz = K-split); grid is 1-D with nx*ny*nz workgroups
__device__ __forceinline__ void tile_coords(const int nx, const int ny, const int swz, int &tx, int &ty, int &tz) {
    const unsigned nwg = gridDim.x, id = blockIdx.x;
    unsigned L = id;
    if (swz) {
        const unsigned q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    if (swz & 2) {
        // grouped order (dense GEMM with many M-tiles, the FC weight gradients): groups of 8 M-tiles, inside a group N-tile major,
        // M-tile fastest -- an XCD's run of ids walks ONE group's 8 A panels (2 MB, L2-resident) across the N-tiles, so that the B
        // panel of a column is fetched once per group instead of all of A once per column (FC6: 32 M-tiles = 8 MB of dY re-read for
        // each of 196 columns)
        const unsigned per = (unsigned)nx * ny, z = L / per, r = L - z * per;
        const unsigned grp = r / (8u * nx), first = grp * 8u, gsz = min((unsigned)ny - first, 8u), rr = r - grp * 8u * nx;
        tz = (int)z;
        tx = (int)(rr / gsz);
        ty = (int)(first + rr - (unsigned)tx * gsz);
        return;
    }
    const unsigned t = L / (unsigned)ny;
    ty = (int)(L - t * ny);
    tz = (int)(t / (unsigned)nx);
    tx = (int)(t - (unsigned)tz * nx);
}

struct ConvGeom {
    // tensor the B operand gathers from: [batch, CB, HB, WB]
    int batch, CB, HB, WB;
    // pixel grid that indexes N (fwd: output OHxOW; dgrad: input IHxIW)
    int PH, PW;
    int pad;
    int M, N, K;
    int k_per_split;  // multiple of BK
    int nx, ny, swz;  // tile grid (N-tiles, M-tiles) and the XCD swizzle switch, see tile_coords
    int slab_aligned; // CB % BK == 0: a K-slab is one filter tap, K is channel-block major, weights are packed [K][mpad]
                      // and the direct-to-LDS kernel runs; otherwise tap-major [M][K] weights and the gather kernel
    int mpad;         // M rounded up to the M-tile (leading dimension of the [K][mpad] packed weights)
    int row_period;   // > 0 (stride 1 only): the image is a STACK of independent maps of row_period rows each (the channel-major
                      // RoI-head layout [1, C, R*7, 7]): a tap is valid when it stays inside its own map -- vertical validity is
                      // tested on py % row_period against row_period instead of py against the image height.  0: plain image.
    const float *zp;  // zero page: source of every out-of-range gather lane
    Div dPHW, dPW, dCB;
    // Stride-2 data gradient, PARITY classes (direct-to-LDS kernel, even PH / PW).  Tap (kh, kw) contributes to input pixel (py, px)
    // only where py + pad - kh and px + pad - kw are even: with the pixels in their natural order every tile holds all four
    // (py & 1, px & 1) classes, so every one of the KH x KW taps has to be walked although three quarters of the products are zero.
    // parity = 1 renumbers the N axis class by class -- n = cls * ncp + (index inside the class, padded to whole tiles), cls =
    // (py & 1) * 2 + (px & 1) -- so that a tile belongs to ONE class and its K loop visits only that class's taps (1 + 2 + 2 + 4
    // of the 9 for 3x3: a quarter of the slabs).  Same products, same order per output element.
    int parity, nc, ncp;   // pixels per class (batch * PH/2 * PW/2), the same rounded up to whole tiles
    Div dNCP, dPQ, dPQW;   // ncp; PH/2 * PW/2; PW/2
};

// pixel of logical N index n: false if n is padding.  PP: the instantiation can run class-major at all (stride-2 data gradient);
// every other instantiation keeps exactly the natural-order code (the 50-us decoder launches are sensitive to every extra
// instruction of their prologue / staging stream: +11 % when this was a run-time branch for all of them).
template <bool PP>
__device__ __forceinline__ bool conv_n_to_pixel(const ConvGeom &g, const int n, int &img, int &py, int &px) {
    if (!PP || !g.parity) {
        if (n >= g.N) { img = py = px = 0; return false; }
        int pix;
        g.dPHW.divmod(n, img, pix);
        g.dPW.divmod(pix, py, px);
        return true;
    }
    int cls, c, rem, i, j;
    g.dNCP.divmod(n, cls, c);
    if (c >= g.nc) { img = py = px = 0; return false; }
    g.dPQ.divmod(c, img, rem);
    g.dPQW.divmod(rem, i, j);
    py = 2 * i + (cls >> 1);
    px = 2 * j + (cls & 1);
    return true;
}

struct Epi {
    float *out;          // final destination (when splits == 1)
    float *ws;           // split-K slabs [splits][M][N] (when splits > 1)
    const float *bias;   // per-M (conv) or per-N (dense); may be null
    int bias_on_n;
    int act;
    float slope;
    int splits;
    int accumulate;      // splits == 1 only: out += result (one writer per element)
    // data gradient only: act'() of the PRODUCER of this conv's input, applied here so that the producer's backward does not need
    // its own elementwise pass.  mask_src = this conv's input x (the producer's ReLU / LeakyReLU output, same shape as the result);
    // result *= x > 0 ? 1 : mask_slope (0 for ReLU).  null: no masking.
    const float *mask_src;
    float mask_slope;
};

// ---------------------------------------------------------------------------
// conv forward / dgrad
// ---------------------------------------------------------------------------
// offset of filter tap (kh,kw) for pixel (py,px) inside one channel plane of the gathered tensor, or -1 if the tap falls
// into the padding / between the strides (dgrad) / the pixel itself is out of range
template <int S, bool DGRAD>
__device__ __forceinline__ int conv_tap_offset(const ConvGeom &g, const bool n_ok, const int py, const int px, const int kh,
                                               const int kw) {
    if (!n_ok) return -1;
    if (S == 1 && g.row_period) {   // stacked maps (see ConvGeom::row_period): same-size convolution, both directions alike
        const int dy = DGRAD ? g.pad - kh : kh - g.pad, dx = DGRAD ? g.pad - kw : kw - g.pad;
        const int my = py % g.row_period + dy, ix = px + dx;
        return ((unsigned)my < (unsigned)g.row_period && (unsigned)ix < (unsigned)g.WB) ? (py + dy) * g.WB + ix : -1;
    }
    if (!DGRAD) {
        const int iy = py * S + kh - g.pad, ix = px * S + kw - g.pad;
        return ((unsigned)iy < (unsigned)g.HB && (unsigned)ix < (unsigned)g.WB) ? iy * g.WB + ix : -1;
    } else {
        const int ty = py + g.pad - kh, tx = px + g.pad - kw;
        if (ty < 0 || tx < 0) return -1;
        const int oy = ty / S, ox = tx / S;
        return (oy * S == ty && ox * S == tx && oy < g.HB && ox < g.WB) ? oy * g.WB + ox : -1;
    }
}

// accumulators -> split-K slab, or (+bias) -> activation -> NCHW output; lanes run along N (pixels): 128-byte row segments.
// Two passes per 32 x 32 block -- first every load the block needs (its 16 bias values once per row block, its 16 mask values),
// then arithmetic and stores.  Written element by element ("load bias, add, activate, load mask, store") the compiler cannot
// batch anything: the stores may alias the loads as far as it knows, so every element waited for its own loads with vmcnt(0) --
// 64 round trips to L2 per wave at the end of every workgroup, with the matrix pipes idle (conv1_2 forward: 66 of 369 us).
template <int WM, int WN, bool PP = false>
__device__ __forceinline__ void conv_epilogue(f32x16 (&acc)[WM / 32][WN / 32], const ConvGeom &g, const Epi &e, const int m0,
                                              const int n0, const int tz, const int wm, const int wn, const int lane) {
    constexpr int TM = WM / 32, TN = WN / 32;
    const int lr = lane & 31;
    const float *__restrict__ bias = e.bias;
    const float *__restrict__ msk = e.mask_src;
    float *__restrict__ out = e.out;
    float *__restrict__ ws = e.ws;
    const size_t plane = (size_t)g.dPHW.d;
    bool ok[TN];
    int nn[TN];
    size_t pbase[TN];      // offset of (image, channel 0, pixel) in the NCHW output
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int n = n0 + wn * WN + j * 32 + lr;
        int oimg = 0, opix = 0;
        if (PP && g.parity) {
            int opy, opx;
            ok[j] = conv_n_to_pixel<true>(g, n, oimg, opy, opx);
            opix = opy * g.PW + opx;
            n = oimg * g.dPHW.d + opix;          // natural index of the pixel (the logical one is class-major)
        } else {
            ok[j] = n < g.N;
            g.dPHW.divmod(ok[j] ? n : 0, oimg, opix);
        }
        nn[j] = n;
        pbase[j] = (size_t)oimg * g.M * plane + opix;
    }
    const int mrow = m0 + wm * WM + 4 * (lane >> 5);     // frag_row(r, lane) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    if (e.splits > 1) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (!ok[j]) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mrow + i * 32 + (r & 3) + 8 * (r >> 2);
                    if (m >= g.M) continue;
                    ws[((size_t)tz * g.M + m) * g.N + nn[j]] = acc[i][j][r];
                }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = 0.f;
        if (bias) {
#pragma unroll
            for (int r = 0; r < 16; ++r) bv[r] = bias[min(mrow + i * 32 + (r & 3) + 8 * (r >> 2), g.M - 1)];
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (!ok[j]) continue;
            float mv[16];
            if (msk) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = min(mrow + i * 32 + (r & 3) + 8 * (r >> 2), g.M - 1);
                    mv[r] = msk[pbase[j] + (size_t)m * plane];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mrow + i * 32 + (r & 3) + 8 * (r >> 2);
                if (m >= g.M) continue;
                float v = acc[i][j][r];
                if (bias) v += bv[r];
                v = apply_act(v, e.act, e.slope);
                if (msk) v = mv[r] > 0.f ? v : v * e.mask_slope;
                out[pbase[j] + (size_t)m * plane] = v;
            }
        }
    }
}

// ---- (1) generic gather kernel: any channel count; weights [M][K] tap-major, operands staged through registers ----------
// Used by the layers whose reduced channel count is not a multiple of 16 (the 3-channel stems and heads).  Gathers are
// issued UNCONDITIONALLY from a pointer that was selected BEFORE the load (zero page for padding / out-of-range lanes): a
// predicated `cond ? *p : 0` made hipcc wrap every single load in its own exec-mask branch.
template <int BM, int BN, int KH, int KW, int S, bool DGRAD>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const float *__restrict__ Wm, const float *__restrict__ X,
                                                         const ConvGeom g, const Epi e) {
    using T = TileCfg<BM, BN>;
    __shared__ __attribute__((aligned(16))) float lds[T::LDS_FLOATS];
    auto As = [&](int b) -> float * { return lds + b * (BK * T::LDA); };
    auto Bs = [&](int b) -> float * { return lds + 2 * BK * T::LDA + b * (BK * T::LDB); };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int tx, ty, tz;
    tile_coords(g.nx, g.ny, g.swz, tx, ty, tz);
    const int m0 = ty * BM, n0 = tx * BN;
    const int k_begin = tz * g.k_per_split;
    const int k_end = min(g.K, k_begin + g.k_per_split);

    const int ka = tid & 15, ra = tid >> 4;             // A: lanes along K, rows tid/16 + 16*j
    constexpr int KS = 256 / BN;                        // B: lanes along N (pixels), k = tid/BN + KS*j (wave-uniform)
    const int nb = tid % BN, kb = tid / BN;

    const int n_glob = n0 + nb;
    const bool n_ok = n_glob < g.N;
    int img, pix, py, px;
    g.dPHW.divmod(n_ok ? n_glob : 0, img, pix);
    g.dPW.divmod(pix, py, px);
    const float *xb = X + (size_t)img * g.CB * g.HB * g.WB;
    const int plane = g.HB * g.WB;

    float ar[T::A_ELEMS], br[T::B_ELEMS];
    auto tap_offset = [&](int kh, int kw) -> int { return conv_tap_offset<S, DGRAD>(g, n_ok, py, px, kh, kw); };

    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < T::A_ELEMS; ++j) {
            const int m = m0 + ra + 16 * j, k = k0 + ka;
            const float *pa = (m < g.M && k < k_end) ? Wm + (size_t)m * g.K + k : g.zp;
            ar[j] = *pa;
        }
#pragma unroll
        for (int j = 0; j < T::B_ELEMS; ++j) {
            const int k = k0 + kb + KS * j;  // wave-uniform
            const bool k_ok = k < k_end;
            const int kk = k_ok ? k : 0;
            const int r = g.dCB.div(kk), c = kk - r * g.CB;
            const int kh = r / KW, kw = r - kh * KW;
            const int off = k_ok ? tap_offset(kh, kw) : -1;
            const float *pb = off >= 0 ? xb + (size_t)c * plane + off : g.zp;
            br[j] = *pb;
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < T::A_ELEMS; ++j) As(buf)[ka * T::LDA + ra + 16 * j] = ar[j];
#pragma unroll
        for (int j = 0; j < T::B_ELEMS; ++j) Bs(buf)[(kb + KS * j) * T::LDB + nb] = br[j];
    };

    f32x16 acc[T::TM][T::TN];
    zero_acc<BM, BN>(acc);

    gload(k_begin);
    sstore(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        const bool more = k0 + BK < k_end;
        if (more) gload(k0 + BK);  // global loads in flight under the MFMAs
        mma_slab<BM, BN>(As(buf), Bs(buf), acc, wm, wn, lane);
        if (more) sstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    conv_epilogue<BM / 2, BN / 2>(acc, g, e, m0, n0, tz, wm, wn, lane);
}

// ---- (2) direct-to-LDS kernel: channel count % 16 == 0 (every VGG / RPN / decoder / discriminator body layer) -----------
// Both operands go global -> LDS with global_load_lds (no VGPR round trip, no ds_write pass):
//   A  weights packed [K][mpad], M contiguous: one dwordx4 instruction lays down 256 consecutive floats = 256/BM k-rows
//   B  one dword instruction = 64 consecutive pixels of one (tap, channel) row; each lane owns ONE pixel for all of its
//      rows, so the tap decode / bounds test is done once per slab and the per-row address is base + j*plane; padding and
//      out-of-range lanes read the zero page (pointer selected before the load)
// LDS is a ring of NST = 4 stages of [16][BM] + [16][BN] floats (rows unpadded: the LDS-DMA destination is
// wave-uniform base + lane*size; the MFMA operand fetch reads 32 consecutive words per half-wave, conflict-free).
// Schedule per K-slab s: issue the loads of slab s+2, wait until this wave's loads of slab s have landed
// (s_waitcnt vmcnt(2L), counted -- never 0 in the steady state), ONE s_barrier, MFMAs of slab s out of stage s%4 with the
// operand fragments double-buffered in registers (LDS reads of K-pair p+1 issued before the MFMAs of K-pair p).
//   WAR: stage (s+2)%4 was last read by the MFMAs of slab s-2; every wave finished those before arriving at barrier s-1.
//   RAW: each wave counts its own LDS-DMA down, the barrier then orders every wave's data before any ds_read.
// Measured against the register-staged kernel (scripts/ablate/conv_glds.hip): +10 % conv2_2, +16 % conv3_2, +21 % conv4_2.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
#define SCDA_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// LDS-DMA through a buffer descriptor over [base, base + 2 GB): lane offsets with bit 31 set are out of range, and an
// out-of-range lane writes 0.0 to its LDS slot -- the hardware's bounds check takes the place of a zero page and of the
// 64-bit address select per lane.  (The descriptor type only exists in the device compilation.)
#if defined(__HIP_DEVICE_COMPILE__)
#define SCDA_BUFFER_LOAD_LDS(BYTES_)                                                                                      \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)0x80000000u, 0x00020000), \
                                             (lds_void_t *)lds_dst, BYTES_, voffset, soffset, 0, 0)
#else
#define SCDA_BUFFER_LOAD_LDS(BYTES_) (void)0
#endif
// address = base + voffset (per lane) + soffset (scalar, < 2^31 - the largest in-range voffset)
__device__ __forceinline__ void buffer_load_lds_b32(const void *base, const unsigned voffset, float *lds_dst, const int soffset = 0) { SCDA_BUFFER_LOAD_LDS(4); }
__device__ __forceinline__ void buffer_load_lds_b128(const void *base, const unsigned voffset, float *lds_dst, const int soffset = 0) { SCDA_BUFFER_LOAD_LDS(16); }
#undef SCDA_BUFFER_LOAD_LDS

// ---- operand layouts in LDS shared by the dense GEMM and the weight-gradient kernels (fragment readers) ------------------
// They use the ring / staging-wave / barrier schedule of conv_igemm_glds_kernel (below); what differs is how a tile lands in
// LDS (the staging side is GldsStager for the GEMM, the staging-wave branch of conv_wgrad_glds_kernel for the gradient):
//   MC (stored [K][MN], MN contiguous): rows of the tile are K-rows, laid down as [16][BMN] by dwordx4 LDS-DMA; the MFMA
//      operand fetch is one ds_read_b32 per K value (32 consecutive words per half-wave).
//   KC (stored [MN][K], K contiguous -- activations and nn.Linear weights): a dwordx4 covers 4 consecutive K of one row, 4
//      lanes cover the 16-deep slab of that row, so the tile lands as [BMN][16].  The MFMA wants, per lane, one value per K
//      step; lane (i = lane&31, h = lane>>5) therefore takes K = 8q + 4h + t (q = 0,1; t = 0..3) -- FOUR consecutive K in one
//      ds_read_b128 -- and MFMA (q,t) contracts K = {8q+t, 8q+4+t}.  (Any pairing is valid as long as A and B agree; MC
//      operands simply read rows 8q+4h+t.)  16-byte pieces of a row are XOR-swizzled with row bits 2..3 so that the 16
//      lanes of one b128 pass hit 16 distinct 16-byte slots of the 256-byte bank row; the LDS-DMA destination is linear, so
//      the permutation is applied to the per-lane SOURCE address (chunk c of row r is fetched by the lane whose slot is
//      c ^ ((r>>2)&3)) and undone by the reader.
template <int BMN, bool MC, int NW = 4>
struct GldsOperand {
    // fragment for K-group q of the 32-row MFMA tile starting at tile row `r0`: f[t] = operand[r0 + lr][8q + 4h + t]
    __device__ static __forceinline__ void frag(const float *stage, const int r0, const int lr, const int h, const int q,
                                                float (&f)[4]) {
        if (MC) {
#pragma unroll
            for (int t = 0; t < 4; ++t) f[t] = stage[(8 * q + 4 * h + t) * BMN + r0 + lr];
        } else {
            const int row = r0 + lr, p = (2 * q + h) ^ ((row >> 2) & 3);
            // a clang vector of floats, NOT HIP's float4 struct: a struct-typed LDS read loses the float type-based alias
            // info, and the compiler then protects it from the LDS-DMA writes in flight with an s_waitcnt vmcnt(0) right after
            // the barrier -- draining the two prefetched slabs every iteration (the ring's own counted vmcnt already
            // guarantees that the slab being read has landed)
            typedef float f32x4_t __attribute__((ext_vector_type(4)));
            const f32x4_t v = *reinterpret_cast<const f32x4_t *>(stage + row * 16 + 4 * p);
            f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
        }
    }
};


// Tile configuration of the direct-to-LDS conv kernel.
// Compute waves: 2 x 2, except the 64 x 256 tile (layers with <= 64 output channels: conv1_x, the decoder's last stages),
// which is 1 x 4 so that every wave still owns a 64 x 64 sub-tile = 32 MFMAs per barrier, and the 256 x 128 tile, which has
// EIGHT (4 x 2, one workgroup per CU, 96 KB of LDS): the gathered pixel operand is shared by twice as many output channels.
// Staging waves: NP extra waves per workgroup do nothing but issue the LDS-DMA of the ring (they hold no accumulators); the
// compute waves' loop is barrier -> fragment reads -> MFMAs.  With every wave staging its own share, all waves of a
// workgroup went through the ~55-instruction staging phase at the same time (the per-slab barrier keeps them in step) and the
// matrix pipe sat idle for its length; now that phase runs underneath the other waves' MFMAs.  NP is chosen so that two
// slabs of one staging wave's loads fit the 6-bit vmcnt counter (2 L <= 63).
// The 32 x 256 tile (layers with <= 32 output rows: the decoders' 64 -> 32 stage, the data gradient into a 32-channel map): 1 x 4
// compute waves of 32 x 64 -- on a 64-row tile half of every MFMA of such a layer multiplies padding.  Its weight tile is two
// LDS-DMA instructions per slab, fewer than the four staging waves its 64 pixel-row pieces need: staging wave 0 issues both
// (A_P0), so the waves' instruction counts differ and each waits on its own count.
template <int BM, int BN>
struct ConvGldsCfg {
    static constexpr int NWC = BM == 256 ? 8 : 4;                    // compute waves
    static constexpr int WGN = BM == 32 ? 4 : BN == 256 ? 4 : 2, WGM = NWC / WGN;   // ... along N / M
    static constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    static constexpr int A_LPR = BM / 4;            // lanes per weight row (dwordx4 each)
    static constexpr int A_RPI = 64 / A_LPR;        // rows per wave-instruction
    static constexpr int A_TOTAL = BK / A_RPI;      // A instructions per slab
    static constexpr int HALVES = BN / 64;          // 64-pixel pieces per B row
    static constexpr int L_TOTAL = A_TOTAL + BK * HALVES;
    static constexpr int NP = L_TOTAL <= 31 ? 1 : L_TOTAL <= 62 ? 2 : 4;   // staging waves
    static constexpr bool A_P0 = (A_TOTAL % NP) != 0;               // the weight tile is staged by staging wave 0 alone
    static constexpr int A_PP = A_P0 ? A_TOTAL : A_TOTAL / NP, ROWS_PP = BK / NP;    // per staging wave: A instructions, B rows
    static constexpr int LB = ROWS_PP * HALVES;                     // pixel-operand LDS-DMA instructions per staging wave per slab
    static constexpr int L = A_PP + LB;                             // ... all of them (with A_P0: staging wave 0's count)
    static constexpr int THREADS = (NWC + NP) * 64;
    // waves per SIMD the register allocation must leave room for: the 64 x 256 tile is 8 waves and two workgroups share a CU
    // (80 KB of LDS each) -- 4 per SIMD, 128 registers (the batched epilogue loads would otherwise take it to 132 and one workgroup)
    static constexpr int WAVES_PER_EU = (BM == 64 && BN == 256) ? 4 : 1;
    static_assert(BK % NP == 0 && 2 * L <= 63 && WM >= 32 && WN >= 32, "staging split");
};

template <int BM, int BN, int KH, int KW, int S, bool DGRAD>
__global__ __launch_bounds__((ConvGldsCfg<BM, BN>::THREADS), (ConvGldsCfg<BM, BN>::WAVES_PER_EU)) void conv_igemm_glds_kernel(const float *__restrict__ Wt,
                                                                                     const float *__restrict__ X,
                                                                                     const ConvGeom g, const Epi e) {
    using C = ConvGldsCfg<BM, BN>;
    constexpr int NST = 4, STAGE = BK * (BM + BN);
    constexpr int NWC = C::NWC, WGN = C::WGN, WM = C::WM, WN = C::WN, TM = C::TM, TN = C::TN;
    constexpr int A_LPR = C::A_LPR, A_RPI = C::A_RPI, HALVES = C::HALVES, L = C::L;
    __shared__ __attribute__((aligned(16))) float lds[NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tx, ty, tz;
    tile_coords(g.nx, g.ny, g.swz, tx, ty, tz);
    const int m0 = ty * BM, n0 = tx * BN;
    // parity classes (ConvGeom::parity): this tile's class and its taps kh = k0h + 2 i, kw = k0w + 2 j; its K loop has
    // CB/16 * nt slabs, split evenly.  Otherwise the K-slab range of this split.
    int k0h = 0, k0w = 0, ntw = KW, nt = KH * KW;
    int s_begin = tz * (g.k_per_split / BK);
    int s_end = min(g.K, (tz + 1) * g.k_per_split) / BK;
    if (S == 2 && DGRAD && g.parity) {
        const int cls = __builtin_amdgcn_readfirstlane(g.dNCP.div(n0));
        k0h = ((cls >> 1) + g.pad) & 1; k0w = ((cls & 1) + g.pad) & 1;
        ntw = (KW - k0w + 1) >> 1;
        nt = ((KH - k0h + 1) >> 1) * ntw;
        const int slabs = (g.CB / BK) * nt, per = (slabs + e.splits - 1) / e.splits;
        s_begin = min(slabs, tz * per);
        s_end = min(slabs, s_begin + per);
    }

    if (wave >= NWC) {
        // ---- staging wave p: A instructions [p*A_PP, (p+1)*A_PP), B rows [p*ROWS_PP, (p+1)*ROWS_PP) of every slab ----------
        // Loads go through buffer descriptors (buffer_load_lds_*): a lane offset with bit 31 set is out of range and lands as
        // 0.0 in LDS, so padding / out-of-range pixels need neither a zero page nor a 64-bit select per load.  A lane's pixels
        // (one per 64-pixel piece of the tile) are fixed for the whole kernel: their offsets inside the gathered tensor
        // (relative to the tile's first image, so that 32 bits are enough -- checked by the launcher) and the validity of each
        // of their KH x KW taps are computed once; a slab then adds one scalar (channel block in the descriptor base, tap
        // offset) and picks its validity bit.
        const int p = wave - NWC;
        // above the compute waves in the SIMD's issue arbitration: at equal priority the MFMA streams of the two compute waves
        // it shares a SIMD with starved this wave and the ring ran dry (conv3_2 forward 121 vs 128 TFLOP/s)
        __builtin_amdgcn_s_setprio(3);
        constexpr unsigned OOB = 0x80000000u;
        // tap offset = lane constant + slab scalar.  Also for the stride-2 data gradient: tap (kh, kw) of input pixel (py, px) reads
        // dY at ((py + pad - kh) / 2, (px + pad - kw) / 2) where both differences are even, and for those taps the quotients are
        // ((py + pad) >> 1) - (kh >> 1) and ((px + pad) >> 1) - (kw >> 1) -- a lane constant minus a slab scalar again; which taps
        // exist for a lane (parity and range) is in its validity bits like the padding.  (The general path re-derived every
        // lane's offset, two divisions included, for every slab: the discriminators' stride-2 data gradients ran at 11-25 TFLOP/s.)
        constexpr bool FAST = KH * KW <= 32 && S <= 2;
        const int plane = g.HB * g.WB;
        constexpr bool PP = S == 2 && DGRAD;
        int img_first;
        if (PP && g.parity) {
            int fy, fx;
            conv_n_to_pixel<true>(g, n0, img_first, fy, fx);     // a tile's first pixel is never padding
            img_first = __builtin_amdgcn_readfirstlane(img_first);
        } else {
            img_first = __builtin_amdgcn_readfirstlane(g.dPHW.div(n0));
        }
        const char *xbase = reinterpret_cast<const char *>(X + ((size_t)img_first * g.CB + p * C::ROWS_PP) * plane);
        int lane_img[HALVES], lane_base[HALVES], py[HALVES], px[HALVES];
        unsigned off_taps[HALVES];   // bit r set: tap r of the lane's pixel reads padding (or the pixel is beyond N)
        bool n_ok[HALVES];
#pragma unroll
        for (int h = 0; h < HALVES; ++h) {
            int img;
            if (PP && g.parity) {
                n_ok[h] = conv_n_to_pixel<true>(g, n0 + h * 64 + lane, img, py[h], px[h]);
                if (!n_ok[h]) img = img_first;
            } else {
                const int n_glob = n0 + h * 64 + lane;
                n_ok[h] = n_glob < g.N;
                int pix;
                g.dPHW.divmod(n_ok[h] ? n_glob : 0, img, pix);
                g.dPW.divmod(pix, py[h], px[h]);
            }
            lane_img[h] = (img - img_first) * g.CB * plane;
            lane_base[h] = 0; off_taps[h] = 0;
            if (FAST) {
                lane_base[h] = lane_img[h] + (DGRAD ? (S == 2 ? ((py[h] + g.pad) >> 1) * g.WB + ((px[h] + g.pad) >> 1)
                                                                      : (py[h] + g.pad) * g.WB + px[h] + g.pad)
                                                    : (py[h] * S - g.pad) * g.WB + px[h] * S - g.pad);
#pragma unroll
                for (int r = 0; r < KH * KW; ++r)
                    off_taps[h] |= (unsigned)(conv_tap_offset<S, DGRAD>(g, n_ok[h], py[h], px[h], r / KW, r % KW) < 0) << r;
            }
        }
        // A: a lane's 4 consecutive output channels of row i*A_RPI + lane/A_LPR of the slab
        const unsigned a_voff = (unsigned)(((lane / A_LPR) * g.mpad + (lane % A_LPR) * 4) * 4);
        const int a_first = C::A_P0 ? 0 : p * C::A_PP * A_RPI;     // first weight row of the slab this wave stages
        const char *wbase = reinterpret_cast<const char *>(Wt + (size_t)a_first * g.mpad + m0);

        auto issue = [&](int s, int buf) {
            float *Ab = lds + buf * STAGE + a_first * BM;
            float *Bb = lds + buf * STAGE + BK * BM + (p * C::ROWS_PP) * BN;
            int cb, r, kh, kw;
            auto issue_a = [&](const char *wa) {
                if (!C::A_P0 || p == 0) {
#pragma unroll
                    for (int i = 0; i < C::A_PP; ++i) buffer_load_lds_b128(wa, a_voff, Ab + i * A_RPI * BM, i * A_RPI * g.mpad * 4);
                }
            };
            if (S == 2 && DGRAD && g.parity) {      // slab s of this class: channel block s / nt, its (s % nt)-th tap
                cb = s / nt;
                const int ti = s - cb * nt, th = ti / ntw;
                kh = k0h + 2 * th; kw = k0w + 2 * (ti - th * ntw);
                r = kh * KW + kw;
                issue_a(wbase + (size_t)(cb * (KH * KW) + r) * BK * g.mpad * 4);
            } else {                                 // weights first: their address needs no division
                issue_a(wbase + (size_t)s * BK * g.mpad * 4);
                cb = s / (KH * KW); r = s - cb * (KH * KW);
                kh = r / KW; kw = r - kh * KW;
            }
            const char *xs = xbase + (size_t)cb * BK * plane * 4;
#pragma unroll
            for (int h = 0; h < HALVES; ++h) {
                unsigned voff;
                if (FAST) {
                    const int tap = DGRAD ? (S == 2 ? -((kh >> 1) * g.WB + (kw >> 1)) : -(kh * g.WB + kw)) : kh * g.WB + kw;
                    voff = ((unsigned)(lane_base[h] + tap) << 2) | (((off_taps[h] >> r) & 1u) << 31);
                } else {
                    const int off = conv_tap_offset<S, DGRAD>(g, n_ok[h], py[h], px[h], kh, kw);
                    voff = off >= 0 ? (unsigned)(lane_img[h] + off) << 2 : OOB;
                }
#pragma unroll
                for (int j = 0; j < C::ROWS_PP; ++j) buffer_load_lds_b32(xs, voff, Bb + j * BN + h * 64, j * plane * 4);
            }
        };
        if (s_begin < s_end) issue(s_begin, 0);
        if (s_begin + 1 < s_end) issue(s_begin + 1, 1);
        int nbuf = 2;
        for (int s = s_begin; s < s_end; ++s) {
            // slab s+2 goes into the buffer last read in iteration s-2: every compute wave that reached barrier s-1 is done
            // with it.  Then wait until slab s (two slabs back in this wave's queue) has landed, and release it.
            if (s + 2 < s_end) {
                issue(s + 2, nbuf);
                if (C::A_P0 && p != 0) SCDA_WAIT_VMCNT(2 * C::LB); else SCDA_WAIT_VMCNT(2 * L);
            } else if (s + 1 < s_end) {
                if (C::A_P0 && p != 0) SCDA_WAIT_VMCNT(C::LB); else SCDA_WAIT_VMCNT(L);
            } else {
                SCDA_WAIT_VMCNT(0);
            }
            __builtin_amdgcn_s_barrier();
            nbuf = (nbuf + 1) & (NST - 1);
        }
        return;
    }

    // ---- compute waves ------------------------------------------------------------------------------------------------------
    const int wm = wave / WGN, wn = wave % WGN;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lr = lane & 31, lk = lane >> 5;
    int buf = 0;
    for (int s = s_begin; s < s_end; ++s) {
        __builtin_amdgcn_s_barrier();
        const float *ap = lds + buf * STAGE + lk * BM + wm * WM + lr;
        const float *bp = lds + buf * STAGE + BK * BM + lk * BN + wn * WN + lr;
        float a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = ap[i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = bp[j * 32];
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp) {
            const int cur = kp & 1, nxt = cur ^ 1;
            if (kp + 1 < BK / 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[nxt][i] = ap[(2 * kp + 2) * BM + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[nxt][j] = bp[(2 * kp + 2) * BN + j * 32];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
        }
        buf = (buf + 1) & (NST - 1);
    }
    conv_epilogue<WM, WN, (S == 2 && DGRAD)>(acc, g, e, m0, n0, tz, wm, wn, lane);
}

// split-K reduce for conv outputs: fixed summation order s = 0..splits-1
// VEC: four consecutive pixels per thread with 16-byte loads / stores (N and the plane size multiples of 4: a group never straddles
// two output channels or two images); same additions in the same order per element as the scalar form.
template <bool VEC>
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const float *__restrict__ ws, const int splits,
                                                                 const int M, const int N, const Div dPHW,
                                                                 const float *__restrict__ bias, const int act,
                                                                 const float slope, float *__restrict__ out,
                                                                 const float *__restrict__ mask_src, const float mask_slope) {
    const long long total = (long long)M * N;
    if (VEC) {
        typedef float f32x4_t __attribute__((ext_vector_type(4)));
        const long long quads = total >> 2;
        for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < quads; q += (long long)blockDim.x * gridDim.x) {
            const long long idx = q << 2;
            const int m = (int)(idx / N), n = (int)(idx - (long long)m * N);
            f32x4_t v = {0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < splits; ++s) v += *reinterpret_cast<const f32x4_t *>(ws + (size_t)s * total + idx);
            if (bias) v += bias[m];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = apply_act(v[k], act, slope);
            int img, pix;
            dPHW.divmod(n, img, pix);
            const size_t o = ((size_t)img * M + m) * dPHW.d + pix;
            if (mask_src) {
                const f32x4_t ms = *reinterpret_cast<const f32x4_t *>(mask_src + o);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = ms[k] > 0.f ? v[k] : v[k] * mask_slope;
            }
            *reinterpret_cast<f32x4_t *>(out + o) = v;
        }
        return;
    }
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)blockDim.x * gridDim.x) {
        const int m = (int)(idx / N), n = (int)(idx - (long long)m * N);
        float v = 0.f;
        for (int s = 0; s < splits; ++s) v += ws[(size_t)s * total + idx];
        if (bias) v += bias[m];
        v = apply_act(v, act, slope);
        int img, pix;
        dPHW.divmod(n, img, pix);
        const size_t o = ((size_t)img * M + m) * dPHW.d + pix;
        if (mask_src) v = mask_src[o] > 0.f ? v : v * mask_slope;
        out[o] = v;
    }
}

// ---------------------------------------------------------------------------
// conv weight gradient
// ---------------------------------------------------------------------------
struct WgradGeom {
    int batch, Cin, IH, IW, Cout, OH, OW, pad;
    int M, N, K;  // Cout, Cin*KH*KW, batch*OH*OW
    int k_per_split;
    int nx, ny, swz;  // see tile_coords
    int a_vec4;   // OH*OW % 4 == 0 and dY 16-byte aligned: float4 loads of dY
    int row_period;   // see ConvGeom::row_period (stride 1, same-size convolution)
    const float *zp;
    Div dOHW, dOW;
};

// WBK = K-slab depth: 32 pixels = one full 128-byte line per gathered row (the K-contiguous operands of the weight
// gradient are read as [row][16 px] = 64-byte pieces at WBK = 16, i.e. two L1 requests per line)
template <int BM, int BN, int KH, int KW, int S, int WBK>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float *__restrict__ dY, const float *__restrict__ X,
                                                         const WgradGeom g, float *__restrict__ ws) {
    using T = TileCfg<BM, BN, WBK>;
    constexpr int RPP = 256 / WBK;   // rows staged per pass
    __shared__ __attribute__((aligned(16))) float lds[T::LDS_FLOATS];
    auto As = [&](int b) -> float * { return lds + b * (WBK * T::LDA); };
    auto Bs = [&](int b) -> float * { return lds + 2 * WBK * T::LDA + b * (WBK * T::LDB); };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int tx, ty, tz;
    tile_coords(g.nx, g.ny, g.swz, tx, ty, tz);
    const int m0 = ty * BM, n0 = tx * BN;
    const int k_begin = tz * g.k_per_split;
    const int k_end = min(g.K, k_begin + g.k_per_split);

    // both operands are contiguous along K (= output pixels): lanes along K
    //   A (dY rows = co): vec4 path lane -> (row = tid/4 [+64], 4 consecutive pixels); scalar path k = tid%16, rows tid/16+16j
    //   B (X gather, rows n = (ci,kh,kw)): k = tid%16, rows tid/16 + 16*j
    const int kl = tid % WBK, rl = tid / WBK;
    constexpr int QPR = WBK / 4;      // float4 pieces per row
    const int qa = tid % QPR, rva = tid / QPR;
    constexpr int RPV = 256 / QPR;    // rows per vec4 pass
    const int ohw = g.dOHW.d, ihw = g.IH * g.IW;

    // per-row constants of the B gather: n -> (ci,kh,kw):  signed offset of the tap inside the input plane, and the
    // tap's index into the 9-bit validity mask that is rebuilt once per K-step from 3 row + 3 column tests
    int b_base[T::B_ELEMS], b_tap[T::B_ELEMS];
#pragma unroll
    for (int j = 0; j < T::B_ELEMS; ++j) {
        const int n = n0 + rl + RPP * j;
        const int c = n / (KH * KW), rem = n - c * (KH * KW);
        const int kh = rem / KW, kw = rem - kh * KW;
        b_base[j] = c * ihw + (kh - g.pad) * g.IW + (kw - g.pad);
        b_tap[j] = (n < g.N) ? kh * KW + kw : 31;   // bit 31 of the mask is never set
    }

    float ar[T::A_ELEMS], br[T::B_ELEMS];
    auto gload = [&](int k0) {
        if (g.a_vec4) {
            const int k = k0 + 4 * qa;
            int img, pix;
            g.dOHW.divmod(k < k_end ? k : 0, img, pix);
            const float *dyb = dY + (size_t)img * g.Cout * ohw + pix;
#pragma unroll
            for (int j = 0; j < T::A_ELEMS / 4; ++j) {
                const int m = m0 + rva + RPV * j;
                const float *pa = (k < k_end && m < g.M) ? dyb + (size_t)m * ohw : g.zp;
                const float4 v = *reinterpret_cast<const float4 *>(pa);
                ar[4 * j + 0] = v.x; ar[4 * j + 1] = v.y; ar[4 * j + 2] = v.z; ar[4 * j + 3] = v.w;
            }
        }
        const int k = k0 + kl;
        const bool k_ok = k < k_end;
        int img, pix, oy, ox;
        g.dOHW.divmod(k_ok ? k : 0, img, pix);
        g.dOW.divmod(pix, oy, ox);
        if (!g.a_vec4) {
            const float *dyb = dY + (size_t)img * g.Cout * ohw + pix;
#pragma unroll
            for (int j = 0; j < T::A_ELEMS; ++j) {
                const int m = m0 + rl + RPP * j;
                const float *pa = (k_ok && m < g.M) ? dyb + (size_t)m * ohw : g.zp;
                ar[j] = *pa;
            }
        }
        // validity of each of the KH*KW taps for this output pixel
        unsigned mask = 0;
        if (k_ok) {
#pragma unroll
            for (int kh = 0; kh < KH; ++kh)
#pragma unroll
                for (int kw = 0; kw < KW; ++kw) {
                    const int iy = (g.row_period ? oy % g.row_period : oy * S) + kh - g.pad, ix = ox * S + kw - g.pad;
                    if ((unsigned)iy < (unsigned)(g.row_period ? g.row_period : g.IH) && (unsigned)ix < (unsigned)g.IW) mask |= 1u << (kh * KW + kw);
                }
        }
        const float *xb = X + (size_t)img * g.Cin * ihw + (oy * S) * g.IW + ox * S;
#pragma unroll
        for (int j = 0; j < T::B_ELEMS; ++j) {
            const float *pb = ((mask >> b_tap[j]) & 1u) ? xb + b_base[j] : g.zp;
            br[j] = *pb;
        }
    };
    auto sstore = [&](int buf) {
        if (g.a_vec4) {
#pragma unroll
            for (int j = 0; j < T::A_ELEMS / 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) As(buf)[(4 * qa + i) * T::LDA + rva + RPV * j] = ar[4 * j + i];
        } else {
#pragma unroll
            for (int j = 0; j < T::A_ELEMS; ++j) As(buf)[kl * T::LDA + rl + RPP * j] = ar[j];
        }
#pragma unroll
        for (int j = 0; j < T::B_ELEMS; ++j) Bs(buf)[kl * T::LDB + rl + RPP * j] = br[j];
    };

    f32x16 acc[T::TM][T::TN];
    zero_acc<BM, BN>(acc);
    gload(k_begin);
    sstore(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += WBK) {
        const bool more = k0 + WBK < k_end;
        if (more) gload(k0 + WBK);
        mma_slab<BM, BN, WBK>(As(buf), Bs(buf), acc, wm, wn, lane);
        if (more) sstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    const int lr = lane & 31;
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
        const int n = n0 + wn * T::WN + j * 32 + lr;
        if (n >= g.N) continue;
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * T::WM + i * 32 + frag_row(r, lane);
                if (m < g.M) ws[((size_t)tz * g.M + m) * g.N + n] = acc[i][j][r];
            }
    }
}

// ---- weight gradient, direct-to-LDS ------------------------------------------------------------------------------------
// Both operands are K-contiguous (K = output pixels), i.e. the KC layout of GldsOperand: tiles land as [rows][16 pixels],
// fragments are ds_read_b128 with the K = 8q + 4h + t assignment and the 16-byte XOR swizzle.
//   A = dY rows (output channels): dwordx4 LDS-DMA, 16 rows x 64 bytes per instruction (needs OH*OW % 16 == 0: a slab never
//       straddles two images and stays 16-byte aligned)
//   B = gathered X rows n = (ci, kh, kw): dword LDS-DMA, one instruction = a group of 4 rows x 16 pixels.  The swizzle moves
//       a lane's pixel with the group's class (row>>2)&3, so all groups of one class share the lane -> pixel map: (oy, ox)
//       and the 9-bit tap-validity mask are kept per class.
// Staging waves (see ConvGldsCfg): NP waves issue all LDS-DMA; staging wave p takes A instructions [p*A_PP, (p+1)*A_PP)
// and the row groups of classes [p*CLS_PP, (p+1)*CLS_PP).  Loads go through buffer descriptors (buffer_load_lds_*): the
// range check returns 0 for any lane offset >= 2^31, which replaces the zero page and the 64-bit selects; per-image tensors
// are < 2 GB (checked by the launcher), the image and the slab's first pixel go into the descriptor base (scalar), and a lane's
// 32-bit offset is  A: a constant per instruction (row base + 16-byte chunk after the swizzle, or OOB for rows >= M)
//                   B: pixel offset of the slab (cursor advanced by 16 pixels per slab, no division) + the row's (ci, tap)
//                      constant, with bit 31 set from the tap-validity mask
// The 32-row tile (<= 32 output channels: the decoders' 64 -> 32 stage, whose 64-row tile multiplied 32 rows of padding over 262144
// pixels): compute waves 1 x 4; its dY tile is two LDS-DMA instructions per slab, issued by staging wave 0 alone (A_P0, as in
// ConvGldsCfg).
template <int BM, int BN>
struct WgradGldsCfg {
    static constexpr int NWC = BM == 256 ? 8 : 4;          // compute waves: WGM x WGN
    static constexpr int WGN = BM == 32 ? 4 : 2, WGM = NWC / WGN;
    static constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    static constexpr int A_TOTAL = BM / 16, GROUPS = BN / 4;   // LDS-DMA instructions per slab: dY (16 rows each), X (4 rows each)
    static constexpr int L_TOTAL = A_TOTAL + GROUPS;
    static constexpr int NP = 4;   // one swizzle class per staging wave (measured: 2 waves with two classes each lose 5-15 %)
    static constexpr bool A_P0 = (A_TOTAL % NP) != 0;
    static constexpr int A_PP = A_P0 ? A_TOTAL : A_TOTAL / NP, CLS_PP = 4 / NP, G_PC = GROUPS / 4;   // per staging wave: A instr, classes; groups per class
    static constexpr int LB = CLS_PP * G_PC;
    static constexpr int L = A_PP + LB;
    static constexpr int THREADS = (NWC + NP) * 64;
    static_assert(2 * L <= 63 && WM >= 32 && WN >= 32, "staging split");
};

template <int BM, int BN, int KH, int KW, int S>
__global__ __launch_bounds__((WgradGldsCfg<BM, BN>::THREADS)) void conv_wgrad_glds_kernel(const float *__restrict__ dY,
                                                                                        const float *__restrict__ X,
                                                                                        const WgradGeom g, float *__restrict__ ws,
                                                                                        float *__restrict__ db_ws) {
    using C = WgradGldsCfg<BM, BN>;
    constexpr int NWC = C::NWC;
    using OA = GldsOperand<BM, false, NWC>;   // fragment readers only
    using OB = GldsOperand<BN, false, NWC>;
    constexpr int NST = 4, STAGE = BK * (BM + BN);
    constexpr int WM = C::WM, WN = C::WN, TM = C::TM, TN = C::TN, L = C::L;
    __shared__ __attribute__((aligned(16))) float lds[NST * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tx, ty, tz;
    tile_coords(g.nx, g.ny, g.swz, tx, ty, tz);
    const int m0 = ty * BM, n0 = tx * BN;
    const int s_begin = tz * (g.k_per_split / BK);
    const int s_end = min(g.K, (tz + 1) * g.k_per_split) / BK;
    const int ohw = g.dOHW.d, ihw = g.IH * g.IW;

    if (wave >= NWC) {
        // ---- staging wave ------------------------------------------------------------------------------------------------
        const int p = wave - NWC;
        __builtin_amdgcn_s_setprio(3);
        constexpr unsigned OOB = 0x80000000u;
        const int a_first = C::A_P0 ? 0 : p * C::A_PP;     // first dY instruction of the slab this wave issues
        unsigned a_voff[C::A_PP];
#pragma unroll
        for (int i = 0; i < C::A_PP; ++i) {
            const int row = (a_first + i) * 16 + (lane >> 2);
            const int c = (lane & 3) ^ ((row >> 2) & 3);
            a_voff[i] = m0 + row < g.M ? (unsigned)(((m0 + row) * ohw + 4 * c) * 4) : OOB;
        }
        unsigned b_off[C::CLS_PP][C::G_PC], b_sh[C::CLS_PP][C::G_PC];
        int oy[C::CLS_PP], ox[C::CLS_PP], klp[C::CLS_PP];
        int oyv[C::CLS_PP];   // the row the tap-validity test sees: oy, or oy % row_period for stacked maps (ConvGeom::row_period)
        const int vrows = g.row_period ? g.row_period : g.IH;
        // Stride 1 and OW % 16 == 0 (every VGG / decoder layer of the workload): the 16 pixels of a slab lie in ONE output row.
        // Row validity of the taps and the slab's pixel offset are then scalar (SALU work of the staging wave), a lane's
        // offset is a per-load CONSTANT plus the validity bit, and the per-slab VALU work drops from ~57 to ~28 instructions
        // per class -- this wave's instruction stream is what bounds the kernel (with the address work ablated away it runs at
        // 134 instead of 119 TFLOP/s on conv3_2).
        const bool row_slab = S == 1 && (g.dOW.d % BK) == 0 && !g.row_period;
        const int bias = (g.pad * g.IW + g.pad) * 4;   // bytes the descriptor base is moved back by
        // slab cursor: image and first pixel (scalar), the lane's output pixel per class; advanced by 16 pixels per issue
        int img0, pix0, step_y, step_x;
        g.dOHW.divmod(s_begin * BK, img0, pix0);
        g.dOW.divmod(BK, step_y, step_x);
#pragma unroll
        for (int c = 0; c < C::CLS_PP; ++c) {
            const int cls = p * C::CLS_PP + c;
            const int kl = 4 * (((lane & 15) >> 2) ^ cls) + (lane & 3);   // the lane's pixel inside a slab (after the swizzle)
            g.dOW.divmod(pix0 + kl, oy[c], ox[c]);
            oyv[c] = g.row_period ? oy[c] % g.row_period : oy[c];
#pragma unroll
            for (int j = 0; j < C::G_PC; ++j) {
                const int n = n0 + 16 * j + 4 * cls + (lane >> 4);
                const int ci = n / (KH * KW), rem = n - ci * (KH * KW);
                const int kh = rem / KW, kw = rem - kh * KW;
                b_off[c][j] = n < g.N ? (unsigned)((ci * ihw + (kh - g.pad) * g.IW + (kw - g.pad)) * 4) : OOB;
                b_sh[c][j] = 31 - (kh * KW + kw);
                // row-slab form: + the lane's pixel inside the slab + the bias that keeps every lane offset non-negative
                if (row_slab && n < g.N) b_off[c][j] += (unsigned)(kl * 4 + bias);
            }
            klp[c] = kl - g.pad;
        }
        int oy_s, ox_s;   // row-slab form: output row and first column of the slab (scalar)
        g.dOW.divmod(pix0, oy_s, ox_s);
        const char *a_img = reinterpret_cast<const char *>(dY) + (size_t)img0 * g.Cout * ohw * 4;
        const char *x_img = reinterpret_cast<const char *>(X) + (size_t)img0 * g.Cin * ihw * 4;
        const size_t a_img_stride = (size_t)g.Cout * ohw * 4, x_img_stride = (size_t)g.Cin * ihw * 4;

        auto issue = [&](int buf) {
            float *st = lds + buf * STAGE;
            const char *a_slab = a_img + (size_t)pix0 * 4;
            if (!C::A_P0 || p == 0) {
#pragma unroll
                for (int i = 0; i < C::A_PP; ++i) buffer_load_lds_b128(a_slab, a_voff[i], st + (a_first + i) * 16 * 16);
            }
            float *Bb = st + BK * BM;
            const bool wrap = pix0 + BK >= ohw;
            if (row_slab) {
                unsigned rows = 0;   // copies of the column bits go where the tap rows are inside the image
#pragma unroll
                for (int kh = 0; kh < KH; ++kh) rows |= (unsigned)((unsigned)(oy_s + kh - g.pad) < (unsigned)g.IH) << (kh * KW);
                const int soff = (oy_s * g.IW + ox_s) * 4;
                const char *xb = x_img - bias;
#pragma unroll
                for (int c = 0; c < C::CLS_PP; ++c) {
                    const int cls = p * C::CLS_PP + c;
                    const int ix0 = ox_s + klp[c];
                    unsigned colb = 0;
#pragma unroll
                    for (int kw = 0; kw < KW; ++kw) colb |= (unsigned)((unsigned)(ix0 + kw) < (unsigned)g.IW) << kw;
                    const unsigned off_taps = ~(colb * rows);
#pragma unroll
                    for (int j = 0; j < C::G_PC; ++j)
                        buffer_load_lds_b32(xb, b_off[c][j] | ((off_taps << b_sh[c][j]) & OOB), Bb + (16 * j + 4 * cls) * 16, soff);
                }
                ox_s += BK;
                if (ox_s >= g.dOW.d) { ox_s = 0; ++oy_s; }
                if (wrap) oy_s = 0;
                pix0 += BK;
                if (wrap) { pix0 = 0; a_img += a_img_stride; x_img += x_img_stride; }
                return;
            }
#pragma unroll
            for (int c = 0; c < C::CLS_PP; ++c) {
                const int cls = p * C::CLS_PP + c;
                // validity of the KH x KW taps of the lane's pixel (row bits x column bits), inverted: bit t set = tap t off the image
                unsigned rowb = 0, colb = 0;
#pragma unroll
                for (int kh = 0; kh < KH; ++kh) rowb |= (unsigned)((unsigned)(oyv[c] * S + kh - g.pad) < (unsigned)vrows) << kh;
#pragma unroll
                for (int kw = 0; kw < KW; ++kw) colb |= (unsigned)((unsigned)(ox[c] * S + kw - g.pad) < (unsigned)g.IW) << kw;
                unsigned mask = 0;
#pragma unroll
                for (int kh = 0; kh < KH; ++kh) mask |= ((rowb >> kh) & 1u) ? colb << (kh * KW) : 0u;
                const unsigned off_taps = ~mask;
                const unsigned x_pix = (unsigned)((oy[c] * S * g.IW + ox[c] * S) * 4);
#pragma unroll
                for (int j = 0; j < C::G_PC; ++j)
                    buffer_load_lds_b32(x_img, (x_pix + b_off[c][j]) | ((off_taps << b_sh[c][j]) & OOB), Bb + (16 * j + 4 * cls) * 16);
                // next slab
                ox[c] += step_x; oy[c] += step_y; oyv[c] += step_y;
                if (ox[c] >= g.dOW.d) { ox[c] -= g.dOW.d; ++oy[c]; ++oyv[c]; }
                if (wrap) { oy[c] -= g.OH; if (!g.row_period) oyv[c] -= g.OH; }
                if (g.row_period) while (oyv[c] >= g.row_period) oyv[c] -= g.row_period;   // OH is a multiple of the period
            }
            pix0 += BK;
            if (wrap) { pix0 = 0; a_img += a_img_stride; x_img += x_img_stride; }
        };
        if (s_begin < s_end) issue(0);
        if (s_begin + 1 < s_end) issue(1);
        int nbuf = 2;
        for (int s = s_begin; s < s_end; ++s) {
            if (s + 2 < s_end) {
                issue(nbuf);
                if (C::A_P0 && p != 0) SCDA_WAIT_VMCNT(2 * C::LB); else SCDA_WAIT_VMCNT(2 * L);
            } else if (s + 1 < s_end) {
                if (C::A_P0 && p != 0) SCDA_WAIT_VMCNT(C::LB); else SCDA_WAIT_VMCNT(L);
            } else {
                SCDA_WAIT_VMCNT(0);
            }
            __builtin_amdgcn_s_barrier();
            nbuf = (nbuf + 1) & (NST - 1);
        }
        return;
    }

    // ---- compute waves ------------------------------------------------------------------------------------------------------
    const int wm = wave / C::WGN, wn = wave % C::WGN;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jj = 0; jj < TN; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
    const int lr = lane & 31, lh = lane >> 5;
    // the workgroups of the first N-tile also produce db[m] = sum over pixels of dY[m][.] (one partial per K-split)
    const bool bias_rows = db_ws != nullptr && tx == 0 && wn == 0;
    float rs[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) rs[i] = 0.f;
    int buf = 0;
    // kernel-argument loads still pending at the loop (pointers only the epilogue uses) share lgkmcnt with the LDS reads and
    // return out of order: with one outstanding the compiler can only ever wait for lgkmcnt(0) inside the loop
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    for (int s = s_begin; s < s_end; ++s) {
        __builtin_amdgcn_s_barrier();
        const float *as = lds + buf * STAGE, *bs = as + BK * BM;
        float a[2][TM][4], b[2][TN][4];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int i = 0; i < TM; ++i) OA::frag(as, wm * WM + i * 32, lr, lh, q, a[q][i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) OB::frag(bs, wn * WN + j * 32, lr, lh, q, b[q][j]);
        }
        __builtin_amdgcn_sched_barrier(0);   // all fragment reads in flight before the first MFMA (the scheduler sank K-group 1)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][i][t], b[q][j][t], acc[i][j], 0, 0, 0);
        // fused bias gradient: row sums of dY ride along on the fragments this lane holds anyway.  AFTER the MFMAs: a branch
        // between the reads and the first MFMA made the compiler wait for all eight fragment reads (lgkmcnt(0)) before any
        // matrix work; now the K-group 0 products start while the K-group 1 fragments are still on their way.
        if (bias_rows) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i) rs[i] += (a[q][i][0] + a[q][i][1]) + (a[q][i][2] + a[q][i][3]);
        }
        buf = (buf + 1) & (NST - 1);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + lr;
        if (n >= g.N) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + i * 32 + frag_row(r, lane);
                if (m < g.M) ws[((size_t)tz * g.M + m) * g.N + n] = acc[i][j][r];
            }
    }
    if (bias_rows) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float t = rs[i] + __shfl_xor(rs[i], 32);   // lanes h = 0 and h = 1 hold the two K halves of a row
            const int m = m0 + wm * WM + i * 32 + lr;
            if (lh == 0 && m < g.M) db_ws[(size_t)tz * g.M + m] = t;
        }
    }
}

// out[idx] = (accumulate ? out[idx] : 0) + sum_s ws[s][idx] (+ bias[col]) -> act
// G = split groups per element: a workgroup covers 256/G consecutive elements, thread (g, e) adds the slabs s = g, g+G, ...
// (independent, pipelined loads), the G partial sums are combined in fixed order through LDS.  With one thread per element
// (G = 1) a 150-way split of a small weight matrix was 150 dependent loads on 144 workgroups: 0.8 TB/s, 2.2 ms/iteration.
// VEC: a thread owns FOUR consecutive elements (16-byte loads of every slab, 16-byte read-modify-write of the destination; total %
// 4 == 0, no bias): the same additions in the same order per element, a quarter of the memory instructions -- the weight-gradient
// reduces (14 - 100 slabs of a 0.1 - 2.4 M-element matrix) ran at 1.4 TB/s in the scalar form.
template <int G, bool VEC>
__global__ __launch_bounds__(256) void dense_splitk_reduce_kernel(const float *__restrict__ ws, const int splits,
                                                                  const long long total, const int N,
                                                                  const float *__restrict__ bias, const int bias_on_n,
                                                                  const int act, const float slope,
                                                                  const int accumulate, float *__restrict__ out,
                                                                  const float *__restrict__ db_ws, float *__restrict__ db,
                                                                  const int db_n, const int db_accumulate) {
    constexpr int E = 256 / G;
    constexpr int W = VEC ? 4 : 1;
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    __shared__ float part[G][W * E + 1];
    const int e = threadIdx.x % E, g = threadIdx.x / E;
    for (long long base = (long long)blockIdx.x * (E * W); base < total; base += (long long)gridDim.x * (E * W)) {
        const long long idx = base + (long long)e * W;
        float v[W];
#pragma unroll
        for (int k = 0; k < W; ++k) v[k] = 0.f;
        if (idx < total) {
            if (VEC) {
                f32x4_t a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
                for (int s = g; s < splits; s += G) a += *reinterpret_cast<const f32x4_t *>(ws + (size_t)s * total + idx);
#pragma unroll
                for (int k = 0; k < W; ++k) v[k] = a[k];
            } else {
#pragma unroll 4
                for (int s = g; s < splits; s += G) v[0] += ws[(size_t)s * total + idx];
            }
        }
        if (G > 1) {
#pragma unroll
            for (int k = 0; k < W; ++k) part[g][e * W + k] = v[k];
            __syncthreads();
            if (g == 0) {
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    float t = 0.f;
#pragma unroll
                    for (int i = 0; i < G; ++i) t += part[i][e * W + k];
                    v[k] = t;
                }
            }
        }
        if (g == 0 && idx < total) {
            if (VEC) {
                f32x4_t o = {v[0], v[1], v[2], v[3]};
#pragma unroll
                for (int k = 0; k < W; ++k) o[k] = apply_act(o[k], act, slope);
                f32x4_t *dst = reinterpret_cast<f32x4_t *>(out + idx);
                *dst = accumulate ? *dst + o : o;
            } else {
                float t = v[0];
                if (bias) t += bias_on_n ? bias[idx % N] : bias[idx / N];
                t = apply_act(t, act, slope);
                out[idx] = accumulate ? out[idx] + t : t;
            }
        }
        if (G > 1) __syncthreads();
    }
    if (db_ws) {   // second, tiny job of the same launch: db[m] (+)= sum over splits of the fused bias-gradient partials
        // one workgroup per row, the slabs spread over its threads and combined by a fixed-shape tree (one thread walking
        // 256 slabs was a chain of 256 loads: 70 us for the 32-channel layers, longer than their whole weight reduce)
        float *red = &part[0][0];   // G * (E + 1) >= 256 floats
        for (int m = blockIdx.x; m < db_n; m += gridDim.x) {
            float v = 0.f;
            for (int s = threadIdx.x; s < splits; s += 256) v += db_ws[(size_t)s * db_n + m];
            __syncthreads();
            red[threadIdx.x] = v;
            __syncthreads();
#pragma unroll
            for (int w = 128; w > 0; w >>= 1) {
                if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
                __syncthreads();
            }
            if (threadIdx.x == 0) db[m] = db_accumulate ? db[m] + red[0] : red[0];
        }
    }
}

static int launch_dense_reduce(const float *ws, int splits, long long total, int N, const float *bias, int bias_on_n, int act,
                               float slope, int accumulate, float *out, const float *db_ws, float *db, int db_n,
                               int db_accumulate, hipStream_t st) {
    const bool vec = !bias && (total & 3) == 0 && ((((uintptr_t)ws) | ((uintptr_t)out)) & 15) == 0;
#define DENSE_REDUCE(G_)                                                                                                 \
    do {                                                                                                                 \
        if (vec)                                                                                                         \
            hipLaunchKernelGGL((dense_splitk_reduce_kernel<G_, true>), dim3(ew_grid(total / 4 * G_)), dim3(256), 0, st, ws, splits, total, N, bias, \
                               bias_on_n, act, slope, accumulate, out, db_ws, db, db_n, db_accumulate);                    \
        else                                                                                                             \
            hipLaunchKernelGGL((dense_splitk_reduce_kernel<G_, false>), dim3(ew_grid(total * G_)), dim3(256), 0, st, ws, splits, total, N, bias, \
                               bias_on_n, act, slope, accumulate, out, db_ws, db, db_n, db_accumulate);                    \
    } while (0)
    if (splits >= 32) DENSE_REDUCE(8);
    else if (splits >= 8) DENSE_REDUCE(4);
    else DENSE_REDUCE(1);
#undef DENSE_REDUCE
    return launch_status("dense_splitk_reduce_kernel");
}

// ---------------------------------------------------------------------------
// dense GEMM  C[M][N] = op(A) * op(B)   (FC layers, 1x1 convs handled by conv path)
//   TA = false: A is [M][K] (K contiguous)   TA = true: A is [K][M] (M contiguous)
//   TB = false: B is [N][K] (K contiguous)   TB = true: B is [K][N] (N contiguous)
// ---------------------------------------------------------------------------
struct GemmGeom {
    int M, N, K, lda, ldb, ldc;
    int k_per_split;
    const float *zp;
    int nx, ny, swz;  // see tile_coords
};

template <int BM, int BN, bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                   const GemmGeom g, const Epi e) {
    using T = TileCfg<BM, BN>;
    __shared__ __attribute__((aligned(16))) float lds[T::LDS_FLOATS];
    auto As = [&](int b) -> float * { return lds + b * (BK * T::LDA); };
    auto Bs = [&](int b) -> float * { return lds + 2 * BK * T::LDA + b * (BK * T::LDB); };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int tx, ty, tz;
    tile_coords(g.nx, g.ny, g.swz, tx, ty, tz);
    const int m0 = ty * BM, n0 = tx * BN;
    const int k_begin = tz * g.k_per_split;
    const int k_end = min(g.K, k_begin + g.k_per_split);

    // K-contiguous operand: lanes along K; MN-contiguous operand: lanes along MN
    const int kl = tid & 15, rl = tid >> 4;
    constexpr int KSA = 256 / BM, KSB = 256 / BN;
    const int ma = tid % BM, kma = tid / BM;
    const int nbb = tid % BN, knb = tid / BN;

    float ar[T::A_ELEMS], br[T::B_ELEMS];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < T::A_ELEMS; ++j) {
            if (!TA) {
                const int m = m0 + rl + 16 * j, k = k0 + kl;
                const float *pa = (m < g.M && k < k_end) ? A + (size_t)m * g.lda + k : g.zp;
                ar[j] = *pa;
            } else {
                const int m = m0 + ma, k = k0 + kma + KSA * j;
                const float *pa = (m < g.M && k < k_end) ? A + (size_t)k * g.lda + m : g.zp;
                ar[j] = *pa;
            }
        }
#pragma unroll
        for (int j = 0; j < T::B_ELEMS; ++j) {
            if (!TB) {
                const int n = n0 + rl + 16 * j, k = k0 + kl;
                const float *pb = (n < g.N && k < k_end) ? B + (size_t)n * g.ldb + k : g.zp;
                br[j] = *pb;
            } else {
                const int n = n0 + nbb, k = k0 + knb + KSB * j;
                const float *pb = (n < g.N && k < k_end) ? B + (size_t)k * g.ldb + n : g.zp;
                br[j] = *pb;
            }
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < T::A_ELEMS; ++j) {
            if (!TA) As(buf)[kl * T::LDA + rl + 16 * j] = ar[j];
            else As(buf)[(kma + KSA * j) * T::LDA + ma] = ar[j];
        }
#pragma unroll
        for (int j = 0; j < T::B_ELEMS; ++j) {
            if (!TB) Bs(buf)[kl * T::LDB + rl + 16 * j] = br[j];
            else Bs(buf)[(knb + KSB * j) * T::LDB + nbb] = br[j];
        }
    };

    f32x16 acc[T::TM][T::TN];
    zero_acc<BM, BN>(acc);
    gload(k_begin);
    sstore(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        const bool more = k0 + BK < k_end;
        if (more) gload(k0 + BK);
        mma_slab<BM, BN>(As(buf), Bs(buf), acc, wm, wn, lane);
        if (more) sstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    const int lr = lane & 31;
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
        const int n = n0 + wn * T::WN + j * 32 + lr;
        if (n >= g.N) continue;
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * T::WM + i * 32 + frag_row(r, lane);
                if (m >= g.M) continue;
                float v = acc[i][j][r];
                if (e.splits > 1) {
                    e.ws[((size_t)tz * g.M + m) * g.N + n] = v;
                } else {
                    if (e.bias) v += e.bias_on_n ? e.bias[n] : e.bias[m];
                    v = apply_act(v, e.act, e.slope);
                    float *dst = e.out + (size_t)m * g.ldc + n;
                    *dst = e.accumulate ? *dst + v : v;
                }
            }
    }
}

// ---- dense GEMM, direct-to-LDS ----------------------------------------------------------------------------------------
// One operand's share of the ring for staging wave p of NP: LDS-DMA instructions [p*PP, (p+1)*PP) of the BMN/16 that make a
// 16-deep slab, in the layouts GldsOperand reads.  Everything that does not depend on the slab is fixed at init: the lane
// offset (with the out-of-range rows / columns encoded as bit 31, see buffer_load_lds_*), the scalar step between
// instructions; a slab only moves the descriptor base.
template <int BMN, bool MC, int NP>
struct GldsStager {
    static constexpr int TOTAL = BMN / 16, PP = TOTAL / NP;
    static constexpr int LPR = BMN / 4, RPI = 64 / LPR;   // MC: lanes per K-row, K-rows per instruction
    static_assert(TOTAL % NP == 0, "staging split");
    unsigned voff[MC ? 1 : PP];
    const char *base0;
    long long kstep;   // bytes per unit of K
    int istep;         // bytes between consecutive instructions of this wave
    int dst0;          // first LDS float of this wave's share inside the operand's stage
    __device__ __forceinline__ void init(const float *base, const int ld, const int extent, const int mn0, const int p, const int lane) {
        constexpr unsigned OOB = 0x80000000u;
        if (MC) {   // instruction gi covers K-rows [gi*RPI, +RPI), 4 consecutive MN per lane
            const int col = (lane % LPR) * 4;
            voff[0] = mn0 + col < extent ? (unsigned)(((lane / LPR) * ld + mn0 + col) * 4) : OOB;
            base0 = reinterpret_cast<const char *>(base + (size_t)(p * PP * RPI) * ld);
            kstep = (long long)ld * 4;
            istep = RPI * ld * 4;
            dst0 = p * PP * RPI * BMN;
        } else {    // instruction gi covers MN-rows [gi*16, +16), 4 consecutive K per lane, 16-byte chunks XOR-swizzled
#pragma unroll
            for (int i = 0; i < PP; ++i) {
                const int row = (p * PP + i) * 16 + (lane >> 2);
                const int c = (lane & 3) ^ ((row >> 2) & 3);
                voff[i] = mn0 + row < extent ? (unsigned)(((lane >> 2) * ld + 4 * c) * 4) : OOB;
            }
            base0 = reinterpret_cast<const char *>(base + ((size_t)mn0 + p * PP * 16) * ld);
            kstep = 4;
            istep = 16 * ld * 4;
            dst0 = p * PP * 16 * 16;
        }
    }
    __device__ __forceinline__ void issue(const int k0, float *stage) const {
        const char *b = base0 + (long long)k0 * kstep;
#pragma unroll
        for (int i = 0; i < PP; ++i)
            buffer_load_lds_b128(b, voff[MC ? 0 : i], stage + dst0 + i * (MC ? RPI * BMN : 16 * 16), i * istep);
    }
};

template <int BM, int BN>
struct GemmGldsCfg {
    static constexpr int NWC = BM == 256 ? 8 : 4;          // compute waves: 256 x 128 tile: 4 x 2
    static constexpr int NP = (BM + BN) / 16 > 16 ? 2 : 1; // staging waves (see ConvGldsCfg): at most 16 LDS-DMA instructions each
    static constexpr int L = (BM + BN) / 16 / NP;
    static constexpr int THREADS = (NWC + NP) * 64;
};

template <int BM, int BN, bool TA, bool TB>
__global__ __launch_bounds__((GemmGldsCfg<BM, BN>::THREADS)) void gemm_glds_kernel(const float *__restrict__ A,
                                                                                 const float *__restrict__ B, const GemmGeom g,
                                                                                 const Epi e) {
    using C = GemmGldsCfg<BM, BN>;
    constexpr int NWC = C::NWC, L = C::L;
    using OA = GldsOperand<BM, TA, NWC>;   // fragment readers
    using OB = GldsOperand<BN, TB, NWC>;
    constexpr int NST = 4, STAGE = BK * (BM + BN);
    constexpr int WM = BM / (NWC / 2), WN = BN / 2, TM = WM / 32, TN = WN / 32;
    __shared__ __attribute__((aligned(16))) float lds[NST * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tx, ty, tz;
    tile_coords(g.nx, g.ny, g.swz, tx, ty, tz);
    const int m0 = ty * BM, n0 = tx * BN;
    const int s_begin = tz * (g.k_per_split / BK);
    const int s_end = min(g.K, (tz + 1) * g.k_per_split) / BK;

    if (wave >= NWC) {   // ---- staging wave ------------------------------------------------------------------------------------
        const int p = wave - NWC;
        __builtin_amdgcn_s_setprio(3);
        GldsStager<BM, TA, C::NP> sa;
        GldsStager<BN, TB, C::NP> sb;
        sa.init(A, g.lda, g.M, m0, p, lane);
        sb.init(B, g.ldb, g.N, n0, p, lane);
        auto issue = [&](int s, int buf) {
            float *st = lds + buf * STAGE;
            sa.issue(s * BK, st);
            sb.issue(s * BK, st + BK * BM);
        };
        if (s_begin < s_end) issue(s_begin, 0);
        if (s_begin + 1 < s_end) issue(s_begin + 1, 1);
        int nbuf = 2;
        for (int s = s_begin; s < s_end; ++s) {
            if (s + 2 < s_end) {
                issue(s + 2, nbuf);
                SCDA_WAIT_VMCNT(2 * L);
            } else if (s + 1 < s_end) {
                SCDA_WAIT_VMCNT(L);
            } else {
                SCDA_WAIT_VMCNT(0);
            }
            __builtin_amdgcn_s_barrier();
            nbuf = (nbuf + 1) & (NST - 1);
        }
        return;
    }

    // ---- compute waves ------------------------------------------------------------------------------------------------------
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jj = 0; jj < TN; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
    const int lr = lane & 31, lh = lane >> 5;
    int buf = 0;
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): no kernel-argument load pending inside the loop (see conv_wgrad_glds_kernel)
    for (int s = s_begin; s < s_end; ++s) {
        __builtin_amdgcn_s_barrier();
        const float *as = lds + buf * STAGE, *bs = as + BK * BM;
        float a[2][TM][4], b[2][TN][4];
#pragma unroll
        for (int i = 0; i < TM; ++i) OA::frag(as, wm * WM + i * 32, lr, lh, 0, a[0][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) OB::frag(bs, wn * WN + j * 32, lr, lh, 0, b[0][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i) OA::frag(as, wm * WM + i * 32, lr, lh, 1, a[1][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) OB::frag(bs, wn * WN + j * 32, lr, lh, 1, b[1][j]);
        __builtin_amdgcn_sched_barrier(0);   // all fragment reads in flight before the first MFMA
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][i][t], b[q][j][t], acc[i][j], 0, 0, 0);
        buf = (buf + 1) & (NST - 1);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + lr;
        if (n >= g.N) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + i * 32 + frag_row(r, lane);
                if (m >= g.M) continue;
                float v = acc[i][j][r];
                if (e.splits > 1) {
                    e.ws[((size_t)tz * g.M + m) * g.N + n] = v;
                } else {
                    if (e.bias) v += e.bias_on_n ? e.bias[n] : e.bias[m];
                    v = apply_act(v, e.act, e.slope);
                    float *dst = e.out + (size_t)m * g.ldc + n;
                    *dst = e.accumulate ? *dst + v : v;
                }
            }
    }
}

// ---- dense GEMM on the bf16 matrix pipe with EXACT products ("bf16 x 9") -----------------------------------------------------
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the fp32 MFMA.  An fp32 value is the sum of three bf16 pieces by truncation
// (hi = x & 0xffff0000, mid = (x - hi) & 0xffff0000, lo = x - hi - mid: 8 + 8 + 8 significand bits, both subtractions exact), a
// bf16 x bf16 product is exact in fp32 (16 significand bits), so the nine piece products of a * b add up to its exact 48-bit
// product: nine bf16 MFMAs per 16 k do the multiplications of eight fp32 MFMAs in 9/16 of the matrix-pipe time, and nothing is
// narrower than fp32 anywhere.  What the hardware does with the sums (scripts/micro/bf16x9_probe.hip, profiles/r06_bf16x9_probe.txt):
// the 16 products and C of one instruction are aligned to the largest and TRUNCATED to ~26 bits (2^26 + 15 x 1 - 2^26 -> 8), so
// nine such instructions into one accumulator are slightly worse than the fp32 MFMA's fmaf chain at K = 25088 (rms 3.8e-8 vs 2.8e-8
// of sum|ab|) -- but with the hi x hi products in one accumulator and the eight small products in a second one (2^-8 of the size:
// its truncations do not count) the error is a THIRD of the fp32 MFMA's (max 0.7 - 1.1e-7 vs 2.0 - 2.6e-7, rms 1.0e-8 vs 2.8e-8).
// That is the form here: two accumulator sets, added in the epilogue.
//
// The split costs ~5 VALU operations per value.  Done per wave on its own fragments it is 6.7 VALU per MFMA -- the in-register form
// of the probe runs at 123 (64 x 64 wave tile) / 148 TFLOP/s fp32-equivalent, no faster than the fp32 kernel.  So the split is
// done ONCE PER WORKGROUP through LDS: the fp32 tiles land in a 3-stage ring by LDS-DMA exactly as in gemm_glds_kernel (same
// stagers, same layouts, same staging waves), and while the compute waves run the MFMAs of slab s out of the bf16 image of slab s
// each of them converts 1/8 of slab s + 1 (3 x (4 values -> 3 x 8 bytes) per lane) into the other bf16 image: 60 VALU + 12 LDS
// instructions per 36 MFMAs.  bf16 image: three planes [384 rows][16 k] (32 bytes per row: A rows 0..255, B rows 256..383), the two
// 16-byte halves of a row swapped by s(row) = ((row >> 2) ^ (row >> 3)) & 1 so that the 16 lanes of a ds_read_b128 pass (and the 8 of
// a b128 write) hit distinct banks; a lane's fragment = 8 consecutive k of its row = one ds_read_b128 per piece.
//   iteration s:  wait (own image stores, own share of stage s + 1) | barrier | issue the LDS-DMA of slab s + 3 into ring stage s % 3 |
//                 12 fragment reads of image s % 2 | 36 MFMAs interleaved with the conversion of ring stage (s + 1) % 3 into image (s + 1) % 2
//   (ring stage s % 3 was converted during iteration s - 1, image (s + 1) % 2 was last read during iteration s - 1.)  Eight waves, each
//   computing, converting and staging (3 LDS-DMA instructions per slab): the two accumulator sets + fragments need ~200 registers, which
//   rules out a third wave per SIMD for dedicated staging.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4v_t;

struct GemmX9Cfg {
    static constexpr int BM = 256, BN = 128, ROWS = BM + BN;
    static constexpr int NWC = 8, THREADS = NWC * 64;                 // every wave computes, converts and stages (two accumulator sets +
                                                                      // the fragments need > 168 registers: no third wave per SIMD)
    static constexpr int NSTF = 3, STAGE = BK * ROWS;                 // fp32 ring: floats per stage
    static constexpr int PLANE = ROWS * 32, IMAGE = 3 * PLANE;        // bf16 image: bytes per piece plane / per image
    static constexpr int LDS_BYTES = NSTF * STAGE * 4 + 2 * IMAGE;    // 73728 + 73728
    static constexpr int L = ROWS / 16 / NWC;                         // LDS-DMA instructions per wave and stage (2 of A, 1 of B)
};

__device__ __forceinline__ int x9_swz(const int row) { return ((row >> 2) ^ (row >> 3)) & 1; }

// four consecutive-k fp32 values of one row -> their three bf16 pieces, packed (k ascending), stored at the row's slot
__device__ __forceinline__ void x9_convert_store(const float x0, const float x1, const float x2, const float x3, char *img, const int off) {
    constexpr unsigned SEL = 0x07060302u;     // v_perm_b32: (hi16 of the first operand) << 16 | hi16 of the second
    const float h0 = __uint_as_float(__float_as_uint(x0) & 0xffff0000u), h1 = __uint_as_float(__float_as_uint(x1) & 0xffff0000u);
    const float h2 = __uint_as_float(__float_as_uint(x2) & 0xffff0000u), h3 = __uint_as_float(__float_as_uint(x3) & 0xffff0000u);
    const float r0 = x0 - h0, r1 = x1 - h1, r2 = x2 - h2, r3 = x3 - h3;                        // exact
    const float m0 = __uint_as_float(__float_as_uint(r0) & 0xffff0000u), m1 = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
    const float m2 = __uint_as_float(__float_as_uint(r2) & 0xffff0000u), m3 = __uint_as_float(__float_as_uint(r3) & 0xffff0000u);
    const float l0 = r0 - m0, l1 = r1 - m1, l2 = r2 - m2, l3 = r3 - m3;                        // exact, <= 8 significant bits
    const u32x2_t hp = {__builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), SEL), __builtin_amdgcn_perm(__float_as_uint(x3), __float_as_uint(x2), SEL)};
    const u32x2_t mp = {__builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), SEL), __builtin_amdgcn_perm(__float_as_uint(r3), __float_as_uint(r2), SEL)};
    const u32x2_t lp = {__builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), SEL), __builtin_amdgcn_perm(__float_as_uint(l3), __float_as_uint(l2), SEL)};
    *reinterpret_cast<u32x2_t *>(img + off) = hp;
    *reinterpret_cast<u32x2_t *>(img + GemmX9Cfg::PLANE + off) = mp;
    *reinterpret_cast<u32x2_t *>(img + 2 * GemmX9Cfg::PLANE + off) = lp;
}

#define X9_PIN(x) asm volatile("" : "+v"(x))
// timing builds only (scripts/ablate/x9_ablate.sh; results are wrong by construction): 1 no conversion arithmetic, 2 no image stores,
// 4 fragments read once before the loop, 8 no LDS-DMA inside the loop, 16 no barrier, 32 no ring reads, 64 no MFMAs
#ifndef X9_ABLATE
#define X9_ABLATE 0
#endif
// the conversion of one lane's three items (3 x 4 values) as a program of 66 single operations, so that the K loop can place
// them two at a time behind its MFMAs.  Item t, operation o: 0-3 hi = x & mask (into tmp); 4-7 r = x - hi; 8-9 pack hi; 10-13
// mid = r & mask; 14-17 l = r - mid; 18-19 pack mid; 20-21 pack lo.
struct X9Work {
    float x[3][4], r[3][4], l[3][4];
    float tmp[4];
    u32x2_t hp[3], mp[3], lp[3];
};
__device__ __forceinline__ void x9_op(const int n, X9Work &w) {
    if (X9_ABLATE & 1) return;
    constexpr unsigned SEL = 0x07060302u;     // v_perm_b32: (hi16 of the first operand) << 16 | hi16 of the second
    const int t = n / 22, o = n % 22;
    if (o < 4) { w.tmp[o] = __uint_as_float(__float_as_uint(w.x[t][o]) & 0xffff0000u); X9_PIN(w.tmp[o]); }
    else if (o < 8) { w.r[t][o - 4] = w.x[t][o - 4] - w.tmp[o - 4]; X9_PIN(w.r[t][o - 4]); }
    else if (o < 10) { const int q = o - 8; unsigned v = __builtin_amdgcn_perm(__float_as_uint(w.x[t][2 * q + 1]), __float_as_uint(w.x[t][2 * q]), SEL); X9_PIN(v); w.hp[t][q] = v; }
    else if (o < 14) { w.tmp[o - 10] = __uint_as_float(__float_as_uint(w.r[t][o - 10]) & 0xffff0000u); X9_PIN(w.tmp[o - 10]); }
    else if (o < 18) { w.l[t][o - 14] = w.r[t][o - 14] - w.tmp[o - 14]; X9_PIN(w.l[t][o - 14]); }
    else if (o < 20) { const int q = o - 18; unsigned v = __builtin_amdgcn_perm(__float_as_uint(w.r[t][2 * q + 1]), __float_as_uint(w.r[t][2 * q]), SEL); X9_PIN(v); w.mp[t][q] = v; }
    else { const int q = o - 20; unsigned v = __builtin_amdgcn_perm(__float_as_uint(w.l[t][2 * q + 1]), __float_as_uint(w.l[t][2 * q]), SEL); X9_PIN(v); w.lp[t][q] = v; }
}
__device__ __forceinline__ void x9_store(const X9Work &w, const int t, char *dst) {
    if (X9_ABLATE & 2) return;
    *reinterpret_cast<u32x2_t *>(dst) = w.hp[t];
    *reinterpret_cast<u32x2_t *>(dst + GemmX9Cfg::PLANE) = w.mp[t];
    *reinterpret_cast<u32x2_t *>(dst + 2 * GemmX9Cfg::PLANE) = w.lp[t];
}

// one operand's conversion items of this thread: ITEMS x (row, 4-k chunk).  KC stage image [rows][16] (chunks XOR-swizzled, see
// GldsOperand): thread -> (row = id >> 2, physical slot = id & 3); MC stage image [16][BMN]: thread -> (row = id % BMN, chunk = id / BMN)
template <int BMN, bool MC, int ITEMS>
struct X9Converter {
    int src[ITEMS];    // float offset inside the operand's part of a ring stage
    int dst[ITEMS];    // byte offset inside a piece plane
    __device__ __forceinline__ void init(const int tid, const int row_base) {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int id = tid + 512 * it;
            int row, chunk;
            if (MC) { row = id % BMN; chunk = id / BMN; src[it] = 4 * chunk * BMN + row; }
            else { row = id >> 2; const int slot = id & 3; chunk = slot ^ ((row >> 2) & 3); src[it] = row * 16 + 4 * slot; }
            const int gr = row_base + row;
            dst[it] = gr * 32 + (((chunk >> 1) ^ x9_swz(gr)) * 16) + (chunk & 1) * 8;
        }
    }
    __device__ __forceinline__ void load1(const float *part, const int it, float (&v)[4]) const {
        if (MC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = part[src[it] + j * BMN];
        } else {
            const f32x4v_t q = *reinterpret_cast<const f32x4v_t *>(part + src[it]);
            v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
        }
    }
    __device__ __forceinline__ void load(const float *part, float (*v)[4]) const {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            if (MC) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[it][j] = part[src[it] + j * BMN];
            } else {
                const f32x4v_t q = *reinterpret_cast<const f32x4v_t *>(part + src[it]);
                v[it][0] = q[0]; v[it][1] = q[1]; v[it][2] = q[2]; v[it][3] = q[3];
            }
        }
    }
    __device__ __forceinline__ void store(const float (*v)[4], char *img) const {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) x9_convert_store(v[it][0], v[it][1], v[it][2], v[it][3], img, dst[it]);
    }
};

// Work of a launch = units (tile, slab), tile-major.  SK = false: one workgroup per (tile, K-split) -- the split-K form of the other
// GEMM kernels (partial slabs + dense reduce).  SK = true ("stream-K"): a PERSISTENT grid of G workgroups (one per CU), logical
// workgroup w owns the contiguous unit range [w R, (w + 1) R), R = ceil(units / G): launches whose tile count is not a multiple of the
// CU count lose nothing to the last round (FC6's data gradient: 392 tiles on 256 CUs), and the ring streams ACROSS tile boundaries --
// the next tile's first slabs are in flight while this tile's epilogue stores (the weight gradient's 3136 32-slab tiles).  A range
// cuts at most its first and its last tile: such a partial segment goes to workspace slot (w, 0) if it starts the range, (w, 1)
// otherwise, and gemm_x9_fixup_kernel adds the slots of each cut tile in ascending w (fixed order, no atomics) into C.
struct X9Stream {
    int sk;        // stream-K launch?
    int spt;       // slabs per tile (K / 16)
    int R;         // units per workgroup
    int G;         // persistent workgroups (multiple of 8)
    long long U;   // units
    float *slots;  // [2 G][256][128] partial tiles
};

template <bool TA, bool TB, bool SK>
__global__ __launch_bounds__((GemmX9Cfg::THREADS), 1) void gemm_x9_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                                          const GemmGeom g, const Epi e, const X9Stream sk) {
    using C = GemmX9Cfg;
    constexpr int BM = C::BM, BN = C::BN, NWC = C::NWC, L = C::L, NSTF = C::NSTF, STAGE = C::STAGE;
    __shared__ __attribute__((aligned(16))) float lds[C::LDS_BYTES / 4];
    // the two bf16 images first: every fragment read / image store is then one per-lane base + a 16-bit immediate (image, plane, block)
    char *images = reinterpret_cast<char *>(lds);
    float *ring = lds + 2 * C::IMAGE / 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // this workgroup's units: tile t0 from slab st0 on, nu of them (tile-major: tile = tx * ny + ty, M-tile fastest)
    int t0, st0, nu, wl = 0, tz = 0;
    if (SK) {
        const int id = blockIdx.x;
        wl = (id & 7) * (sk.G >> 3) + (id >> 3);      // XCD x (workgroups x, x + 8, ...) owns a contiguous run of ranges
        const long long u0 = (long long)wl * sk.R, u1 = min(sk.U, u0 + sk.R);
        if (u0 >= u1) return;
        t0 = (int)(u0 / sk.spt); st0 = (int)(u0 - (long long)t0 * sk.spt); nu = (int)(u1 - u0);
    } else {
        int tx, ty;
        tile_coords(g.nx, g.ny, g.swz, tx, ty, tz);
        t0 = tx * g.ny + ty;
        st0 = tz * (g.k_per_split / BK);
        nu = min(g.K, (tz + 1) * g.k_per_split) / BK - st0;
        if (nu <= 0) return;      // (never: every split holds slabs)
    }
    t0 = __builtin_amdgcn_readfirstlane(t0); st0 = __builtin_amdgcn_readfirstlane(st0); nu = __builtin_amdgcn_readfirstlane(nu);
    const int spt = SK ? sk.spt : 0x7fffffff;

    // ---- this wave's share of the fp32 ring (LDS-DMA): issued unconditionally, the cursor stops at the last unit -- behind a branch
    // the compiler's waitcnt pass assumes nothing was issued and drains the prefetch with vmcnt(0); a re-load of the last unit goes
    // into a ring stage nobody reads any more
    GldsStager<BM, TA, NWC> sa;
    GldsStager<BN, TB, NWC> sb;
    int it = t0, is = st0, iu = 0;      // issue cursor: tile, slab, unit
    auto set_tile = [&](int t) {
        // (readfirstlane: the cursor is uniform, but the compiler loses that across the loop-carried updates and then wraps every
        //  LDS-DMA instruction -- whose descriptor must be scalar -- in a waterfall loop)
        t = __builtin_amdgcn_readfirstlane(t);
        const int tx = t / g.ny, ty = t - tx * g.ny;
        sa.init(A, g.lda, g.M, ty * BM, wave, lane);
        sb.init(B, g.ldb, g.N, tx * BN, wave, lane);
    };
    set_tile(it);
    auto advance = [&]() {
        if (iu < nu - 1) {
            ++iu; ++is;
            if (SK && is == spt) { is = 0; ++it; set_tile(it); }
        }
    };
    sa.issue(is * BK, ring); sb.issue(is * BK, ring + BK * BM); advance();
    sa.issue(is * BK, ring + STAGE); sb.issue(is * BK, ring + STAGE + BK * BM); advance();
    sa.issue(is * BK, ring + 2 * STAGE); sb.issue(is * BK, ring + 2 * STAGE + BK * BM); advance();

    const int wm = wave >> 1, wn = wave & 1;
    constexpr int TM = 2, TN = 2;
    f32x16 acc[TM][TN], acs[TM][TN];     // hi x hi | the eight small products
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acs[i][j][r] = 0.f; }
    const int lr = lane & 31, lh = lane >> 5;
    X9Converter<BM, TA, 2> ca;
    X9Converter<BN, TB, 1> cb;
    ca.init(tid, 0);
    cb.init(tid, BM);
    // fragment byte offsets inside a piece plane (row = block base + lr: the swizzle only sees lr)
    const int fa = (wm * 64 + lr) * 32 + ((lh ^ x9_swz(lr)) * 16);
    const int fb = (BM + wn * 64 + lr) * 32 + ((lh ^ x9_swz(lr)) * 16);
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): no kernel-argument load pending inside the loop

    float va[2][4], vb[1][4];
    // iteration -1: convert stage 0 into image 0
    SCDA_WAIT_VMCNT(2 * L);
    __builtin_amdgcn_s_barrier();
    ca.load(ring, va);
    cb.load(ring + BK * BM, vb);
    ca.store(va, images);
    cb.store(vb, images);
    int rbuf = 1, fbuf = 0;   // ring stage of unit s + 1; of unit s (free behind barrier(s))
    int ct = t0, cs = st0;    // compute cursor: tile, slab of unit s
    int seg0 = 0;             // unit at which the current tile's segment began
    u32x4_t a[TM][3], b[TN][3];
#define X9_FA(I_, P_) a[I_][P_] = *reinterpret_cast<const u32x4_t *>(img + (P_) * C::PLANE + fa + (I_) * 32 * 32)
#define X9_FB(J_, P_) b[J_][P_] = *reinterpret_cast<const u32x4_t *>(img + (P_) * C::PLANE + fb + (J_) * 32 * 32)
    if (X9_ABLATE & 4) {
        const char *img = images;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) { X9_FA(0, pc); X9_FA(1, pc); X9_FB(0, pc); X9_FB(1, pc); }
    }
    for (int s = 0; s < nu; ++s) {
        // this wave's image stores are done, its share of stage s + 1 has landed (outstanding: s + 1, s + 2 -- and possibly an
        // epilogue's stores, which are YOUNGER: "at most L outstanding" still implies stage s + 1, loads return in order)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(L) : "memory");
        if (!(X9_ABLATE & 16)) __builtin_amdgcn_s_barrier();
        const char *img = images + (s & 1) * C::IMAGE;
        char *nimg = images + ((s + 1) & 1) * C::IMAGE;
        const float *rs = ring + rbuf * STAGE;      // (the last iteration converts a re-load of the last unit into the image nobody
        X9Work w;                                   //  reads: no branch inside the MFMA stream)
        if (X9_ABLATE) w = X9Work{};
        if (!(X9_ABLATE & 4)) {
        // LDS reads in the order the slots below need them (they return in order): the first MFMA waits for two of them
        X9_FA(0, 0); X9_FB(0, 0);
        __builtin_amdgcn_sched_barrier(0);          // (the scheduler clusters LDS reads by base register otherwise: conversion reads last)
        X9_FB(1, 0);
        if (!(X9_ABLATE & 32)) ca.load1(rs, 0, w.x[0]);      // item 0: this lane's first chunk of A (conversion starts in slot 1)
        X9_FA(1, 0);
        __builtin_amdgcn_sched_barrier(0);
        X9_FA(0, 2); X9_FB(0, 2); X9_FB(1, 2); X9_FA(1, 2);
        X9_FA(0, 1); X9_FB(0, 1); X9_FB(1, 1); X9_FA(1, 1);
        } else if (!(X9_ABLATE & 32)) ca.load1(rs, 0, w.x[0]);
        if (!(X9_ABLATE & 32)) {
            ca.load1(rs, 1, w.x[1]);                // item 1: its second chunk of A
            cb.load1(rs + BK * BM, 0, w.x[2]);      // item 2: its chunk of B
        }
        __builtin_amdgcn_sched_barrier(0);
        // Slot k = MFMA k (product k / 4 on block k % 4: the MFMAs on one accumulator are four instructions apart) + conversion
        // operations 2(k - 1), 2(k - 1) + 1; an item's three image stores in the slot behind its last operation; this wave's three
        // LDS-DMA instructions for unit s + 3 in slots 13 / 25 (underneath the MFMA stream, not in the empty pipe behind the barrier).
        // Written slot by slot and fenced: as blocks (all VALU, then all MFMAs) the two waves of a SIMD stay in phase and the
        // matrix pipe idles for the VALU's length (scripts/micro/bf16x9_probe.hip part 3); left to the scheduler, the MFMAs of
        // one accumulator were bunched into dependent runs.
        // pieces: 0 = hi, 1 = mid, 2 = lo.  hi x hi into acc; the eight small products, smallest first, into acs.
        constexpr int PA[9] = {0, 2, 2, 1, 2, 0, 1, 1, 0}, PB[9] = {0, 2, 1, 2, 0, 2, 1, 0, 1};
        float *fst = ring + fbuf * STAGE;
#pragma unroll
        for (int k = 0; k < 36; ++k) {
            const int pr = k >> 2, i = (k >> 1) & 1, j = k & 1;
            if (X9_ABLATE & 64) {
                X9_PIN(a[i][pr % 3][0]); X9_PIN(b[j][pr % 3][0]);
            } else if (pr == 0) {
                X9_PIN(acc[i][j]);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[i][0]), __builtin_bit_cast(bf16x8_t, b[j][0]), acc[i][j], 0, 0, 0);
                X9_PIN(acc[i][j]);
            } else {
                X9_PIN(acs[i][j]);
                acs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[i][PA[pr]]), __builtin_bit_cast(bf16x8_t, b[j][PB[pr]]), acs[i][j], 0, 0, 0);
                X9_PIN(acs[i][j]);
            }
            if (k >= 1 && 2 * (k - 1) < 66) { x9_op(2 * (k - 1), w); x9_op(2 * (k - 1) + 1, w); }
            if (k == 12) x9_store(w, 0, nimg + ca.dst[0]);
            if (k == 23) x9_store(w, 1, nimg + ca.dst[1]);
            if (k == 34) x9_store(w, 2, nimg + cb.dst[0]);
            if (k == 13 && !(X9_ABLATE & 8)) sa.issue(is * BK, fst);
            if (k == 25 && !(X9_ABLATE & 8)) sb.issue(is * BK, fst + BK * BM);
            __builtin_amdgcn_sched_barrier(0);
        }
        advance();
        rbuf = rbuf == NSTF - 1 ? 0 : rbuf + 1;
        fbuf = fbuf == NSTF - 1 ? 0 : fbuf + 1;
        // ---- end of this tile's segment: epilogue ---------------------------------------------------------------------------
        const bool tile_end = SK && cs == spt - 1;
        if (tile_end || s == nu - 1) {
            // (opaque copies: computed from the loop-invariant lane / wave numbers, the 64 store offsets were hoisted out of the K loop
            //  and held -- i.e. spilled -- across it)
            int lane = tid & 63, lr = lane & 31, wm = wave >> 1, wn = wave & 1;
            X9_PIN(lane); X9_PIN(lr);
            asm volatile("" : "+s"(wm), "+s"(wn));
            const int tx = ct / g.ny, ty = ct - tx * g.ny;
            const int m0 = ty * BM, n0 = tx * BN;
            const bool partial = SK && !(tile_end && cs - (s - seg0) == 0);      // the segment does not cover slabs 0 .. spt - 1
            if (SK && partial) {
                float *slot = sk.slots + ((size_t)2 * wl + (seg0 == 0 ? 0 : 1)) * (BM * BN);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            slot[(wm * 64 + i * 32 + frag_row(r, lane)) * BN + wn * 64 + j * 32 + lr] = acc[i][j][r] + acs[i][j][r];
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn * 64 + j * 32 + lr;
                    if (n >= g.N) continue;
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int m = m0 + wm * 64 + i * 32 + frag_row(r, lane);
                            if (m >= g.M) continue;
                            float v = acc[i][j][r] + acs[i][j][r];
                            if (!SK && e.splits > 1) {
                                e.ws[((size_t)tz * g.M + m) * g.N + n] = v;
                            } else {
                                if (e.bias) v += e.bias_on_n ? e.bias[n] : e.bias[m];
                                v = apply_act(v, e.act, e.slope);
                                float *dst = e.out + (size_t)m * g.ldc + n;
                                *dst = e.accumulate ? *dst + v : v;
                            }
                        }
                }
            }
            if (SK) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acs[i][j][r] = 0.f; }
                seg0 = s + 1;
            }
        }
        if (SK) { if (cs == spt - 1) { cs = 0; ++ct; } else ++cs; }
    }
}

#undef X9_FA
#undef X9_FB

// the cut tiles of a stream-K launch: boundary b (1 .. G - 1) lies inside tile t = b R / spt unless it falls on the tile's first slab;
// the FIRST boundary inside a tile adds up that tile's partial segments, workgroups wa = b - 1 .. wb in ascending order, and applies
// the epilogue.  grid (G - 1, 4): blockIdx.y = a quarter of the tile's rows.
__global__ __launch_bounds__(256) void gemm_x9_fixup_kernel(const GemmGeom g, const Epi e, const X9Stream sk) {
    constexpr int BM = GemmX9Cfg::BM, BN = GemmX9Cfg::BN;
    const int b = blockIdx.x + 1;
    const long long u = (long long)b * sk.R;
    if (u >= sk.U) return;
    const int t = (int)(u / sk.spt);
    const long long F = (long long)t * sk.spt;
    if (u == F || (long long)(b - 1) * sk.R > F) return;      // not inside a tile / not the first boundary inside it
    const int wa = b - 1;
    const int wb = (int)min((F + sk.spt - 1) / sk.R, (long long)sk.G - 1);
    const int slot_a = (long long)wa * sk.R == F ? 0 : 1;
    const int tx = t / g.ny, ty = t - tx * g.ny;
    const int m0 = ty * BM, n0 = tx * BN;
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    for (int idx = threadIdx.x; idx < (BM / 4) * (BN / 4); idx += 256) {
        const int row = blockIdx.y * (BM / 4) + idx / (BN / 4), col = (idx % (BN / 4)) * 4;
        f32x4_t v = *reinterpret_cast<const f32x4_t *>(sk.slots + ((size_t)2 * wa + slot_a) * (BM * BN) + row * BN + col);
        for (int w = wa + 1; w <= wb; ++w) v += *reinterpret_cast<const f32x4_t *>(sk.slots + ((size_t)2 * w) * (BM * BN) + row * BN + col);
        const int m = m0 + row;
        if (m >= g.M) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = n0 + col + q;
            if (n >= g.N) continue;
            float x = v[q];
            if (e.bias) x += e.bias_on_n ? e.bias[n] : e.bias[m];
            x = apply_act(x, e.act, e.slope);
            float *dst = e.out + (size_t)m * g.ldc + n;
            *dst = e.accumulate ? *dst + x : x;
        }
    }
}

// ---- forward of a 3x3, stride-1, pad-1 conv with <= 4 INPUT channels (VGG conv1_1: 3 -> 64 on the full-resolution image) ------
// The GEMM view has K = 27: the gather kernel's two 16-deep slabs are half padding and its register-staged 64 x 64 tiles stream
// the 134 MB result at 2 TB/s (67 us at 512 x 1024).  Direct form: one thread per output pixel (lanes along W: every store of a wave
// is 256 contiguous bytes of one channel row), its 9 x Cin input values in registers, the [Cout][9 * Cin] weights (the gather kernel's
// tap-major packing, as given) read through the scalar unit -- the index is wave-uniform -- as the SGPR operand of the FMAs.  Bound
// by the result's stores.  Same products summed tap-major / channel-minor per output: the gather kernel's K order.
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_small_cin_fwd_kernel(const float *__restrict__ x, const float *__restrict__ wk,
                                                                    const float *__restrict__ bias, float *__restrict__ y, const int H,
                                                                    const int W, const int Cout, const int act, const float slope) {
    const int px = blockIdx.x * 256 + threadIdx.x, py = blockIdx.y, img = blockIdx.z;
    if (px >= W) return;
    const float *xi = x + (size_t)img * CIN * H * W;
    float v[9 * CIN];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int iy = py + kh - 1, ix = px + kw - 1;
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
#pragma unroll
            for (int c = 0; c < CIN; ++c) v[(kh * 3 + kw) * CIN + c] = ok ? xi[((size_t)c * H + iy) * W + ix] : 0.f;
        }
    float *yo = y + ((size_t)img * Cout * H + py) * W + px;
    for (int m = 0; m < Cout; ++m) {
        const float *wm = wk + m * (9 * CIN);
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 9 * CIN; ++k) a = fmaf(wm[k], v[k], a);
        if (bias) a += bias[m];
        yo[(size_t)m * H * W] = apply_act(a, act, slope);
    }
}

// ---- data gradient of a conv with <= 4 INPUT channels (an image-side layer: the discriminators' first conv) -------------
// The implicit GEMM would have M = Cin <= 4 output rows in a 64-row MFMA tile: 95 % of the matrix work on padding (measured
// 188 us for the 3-channel 256x256 batch-4 layer).  This is a direct form instead: one thread per input pixel, the <= 4
// results in registers, the [Cout][Cin][R] weights (unpacked, as stored) in LDS, dY read coalesced along the pixel row --
// bandwidth-bound on dY (17 MB for that layer).  Taps are visited (kh, kw) outer / output channel inner.
template <int KH, int KW>
__global__ __launch_bounds__(256) void conv_dgrad_small_cin_kernel(const float *__restrict__ dY, const float *__restrict__ W,
                                                                   float *__restrict__ dX, const int batch, const int Cin,
                                                                   const int IH, const int IW, const int Cout, const int OH,
                                                                   const int OW, const int S, const int P) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [Cout][R][4]  (Cin padded to 4)
    constexpr int R = KH * KW;
    for (int i = threadIdx.x; i < Cout * R * 4; i += blockDim.x) {
        const int m = i & 3, r = (i >> 2) % R, co = (i >> 2) / R;
        wl[i] = m < Cin ? W[((size_t)co * Cin + m) * R + r] : 0.f;
    }
    __syncthreads();
    const long long total = (long long)batch * IH * IW;
    const int ohw = OH * OW;
    for (long long n = blockIdx.x * (long long)blockDim.x + threadIdx.x; n < total; n += (long long)blockDim.x * gridDim.x) {
        const int ix = (int)(n % IW);
        const long long t = n / IW;
        const int iy = (int)(t % IH), img = (int)(t / IH);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        const float *g = dY + (size_t)img * Cout * ohw;
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
            const int ty = iy + P - kh;
            if (ty < 0 || ty % S != 0 || ty / S >= OH) continue;
#pragma unroll
            for (int kw = 0; kw < KW; ++kw) {
                const int tx = ix + P - kw;
                if (tx < 0 || tx % S != 0 || tx / S >= OW) continue;
                const float *gp = g + (ty / S) * OW + tx / S;
                const float4 *wp = reinterpret_cast<const float4 *>(wl) + (kh * KW + kw);
                for (int co = 0; co < Cout; ++co) {
                    const float v = gp[(size_t)co * ohw];
                    const float4 w4 = wp[co * R];
                    a0 += w4.x * v; a1 += w4.y * v; a2 += w4.z * v; a3 += w4.w * v;
                }
            }
        }
        float *o = dX + ((size_t)img * Cin * IH + iy) * IW + ix;
        const size_t plane = (size_t)IH * IW;
        o[0] = a0;
        if (Cin > 1) o[plane] = a1;
        if (Cin > 2) o[2 * plane] = a2;
        if (Cin > 3) o[3 * plane] = a3;
    }
}

// weight packing for conv_igemm*_kernel:  w[Cout][Cin][R]  ->  the GEMM's A operand; C = the reduced channel dim
//   forward : M = Cout, C = Cin        dgrad : M = Cin, C = Cout
//   C % 16 == 0 : out[K][mpad], k = ((c/16)*R + r)*16 + c%16, columns m >= M zero   (direct-to-LDS kernel)
//   otherwise   : out[M][K],    k = r*C + c                                          (gather kernel)
__host__ __device__ inline int conv_packed_mpad(int M) { return M <= 64 ? 64 : (M + 127) / 128 * 128; }

__global__ __launch_bounds__(256) void pack_weight_kernel(const float *__restrict__ w, float *__restrict__ out,
                                                          const int Cout, const int Cin, const int R,
                                                          const int for_dgrad, const long long total);

// all conv weights of one optimiser group in ONE launch (after its Adam step): desc[j] = {source offset (floats, from
// `base`), destination offset (floats, from `out`), Cout, Cin, R, for_dgrad}, destinations ascending and contiguous
__device__ __forceinline__ float packed_weight_elem(const float *__restrict__ w, const long long idx, const int Cout,
                                                    const int Cin, const int R, const int for_dgrad) {
    const int C = for_dgrad ? Cout : Cin, M = for_dgrad ? Cin : Cout;
    const bool blocked = (C % BK) == 0;
    const int mpad = blocked ? conv_packed_mpad(M) : M;
    const long long K = (long long)C * R;
    int k, m, c, r;
    if (blocked) {
        m = (int)(idx % mpad);
        k = (int)(idx / mpad);
        const int sl = k / BK, cb = sl / R;
        r = sl - cb * R;
        c = cb * BK + (k & (BK - 1));
    } else {
        k = (int)(idx % K);
        m = (int)(idx / K);
        r = k / C;
        c = k - r * C;
    }
    const int co = for_dgrad ? c : m, ci = for_dgrad ? m : c;
    return m < M ? w[((size_t)co * Cin + ci) * R + r] : 0.f;
}

__global__ __launch_bounds__(256) void pack_weight_kernel(const float *__restrict__ w, float *__restrict__ out,
                                                          const int Cout, const int Cin, const int R,
                                                          const int for_dgrad, const long long total) {
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)blockDim.x * gridDim.x)
        out[idx] = packed_weight_elem(w, idx, Cout, Cin, R, for_dgrad);
}

// One workgroup = one tile of one layer.  desc[j] = {source offset (floats, from `base`), destination offset (floats, from `out`),
// Cout, Cin, R, for_dgrad, first tile id of this layer}; the layers' tiles are numbered consecutively.
//   blocked layers (C % 16 == 0, R <= 9): tile = 64 packed columns (m) x one 16-channel block (16*R packed rows).  The tile goes
//     through LDS so that BOTH sides are coalesced: the source is read in runs along its own fastest dimensions ([co][ci][r]:
//     16*R consecutive floats per output channel forward, 64*R per channel for the data-gradient layout), the destination is
//     written in 256-byte row segments along m.  (The element-wise form read the source with a stride of Cin*R floats between
//     neighbouring lanes and divided 64-bit indices per element: 219 us for the detector's 2 x 14.7 M weights, 0.8 TB/s.)
//   other layers (3-channel stems, 30/60-channel heads as C): 4096 elements per tile through packed_weight_elem.
constexpr int PACK_TILE_ELEMS = 4096;
template <int R>
__device__ __forceinline__ void pack_tile_blocked(const float *__restrict__ w, float *__restrict__ out, const int Cout, const int Cin,
                                                  const int for_dgrad, const int t, float *tile) {
    const int C = for_dgrad ? Cout : Cin, M = for_dgrad ? Cin : Cout;
    const int mpad = conv_packed_mpad(M);
    const int tiles_m = mpad / 64;
    const int cb = t / tiles_m, m0 = (t - cb * tiles_m) * 64;
    (void)C;
    constexpr int KL = 16 * R;               // packed rows of the tile
    const int tid = threadIdx.x;
    if (!for_dgrad) {                        // source rows = output channels: 16*R consecutive floats each
        for (int i = tid; i < 64 * KL; i += 256) {
            const int ml = i / KL, j = i - ml * KL;
            const int cl = j / R, r = j - cl * R;
            const float v = (m0 + ml < M) ? w[((size_t)(m0 + ml) * Cin + cb * 16) * R + j] : 0.f;
            tile[(r * 16 + cl) * 65 + ml] = v;
        }
    } else {                                 // source rows = this block's 16 output channels: 64*R consecutive floats each
        for (int i = tid; i < 16 * 64 * R; i += 256) {
            const int cl = i / (64 * R), j = i - cl * (64 * R);
            const int ml = j / R, r = j - ml * R;
            const float v = (m0 + ml < M) ? w[((size_t)(cb * 16 + cl) * Cin + m0) * R + j] : 0.f;
            tile[(r * 16 + cl) * 65 + ml] = v;
        }
    }
    __syncthreads();
    float *o = out + ((size_t)cb * KL) * mpad + m0;
    for (int i = tid; i < KL * 64; i += 256) {
        const int kl = i >> 6, ml = i & 63;
        o[(size_t)kl * mpad + ml] = tile[kl * 65 + ml];
    }
}

__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const float *__restrict__ base, float *__restrict__ out,
                                                                   const long long *__restrict__ desc, const int n) {
    __shared__ float tile[16 * 9 * 65];
    const long long id = blockIdx.x;
    int lo = 0, hi = n - 1;   // last layer whose first tile id is <= id (uniform: scalar loads)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[mid * 7 + 6] <= id) lo = mid; else hi = mid - 1;
    }
    const long long *d = desc + lo * 7;
    const float *w = base + d[0];
    float *o = out + d[1];
    const int Cout = (int)d[2], Cin = (int)d[3], R = (int)d[4], for_dgrad = (int)d[5], t = (int)(id - d[6]);
    if (for_dgrad >= 2) { pack_tile_wino(w, o, Cout, Cin, for_dgrad - 2, t, tile); return; }   // Winograd filters (modes 2 / 3)
    const int C = for_dgrad ? Cout : Cin, M = for_dgrad ? Cin : Cout;
    if ((C % BK) == 0 && R == 9) pack_tile_blocked<9>(w, o, Cout, Cin, for_dgrad, t, tile);
    else if ((C % BK) == 0 && R == 1) pack_tile_blocked<1>(w, o, Cout, Cin, for_dgrad, t, tile);
    else {
        const long long total = (long long)((C % BK) == 0 ? conv_packed_mpad(M) : M) * C * R;
        const long long first = (long long)t * PACK_TILE_ELEMS;
        for (long long idx = first + threadIdx.x; idx < first + PACK_TILE_ELEMS && idx < total; idx += 256)
            o[idx] = packed_weight_elem(w, idx, Cout, Cin, R, for_dgrad);
    }
}

// tiles of one layer in pack_weights_batched_kernel's numbering (the caller builds the descriptor table with it)
SCDA_API long long scda_conv2d_pack_tiles(int Cout, int Cin, int KH, int KW, int for_dgrad) {
    if (for_dgrad >= 2) return (KH == 3 && KW == 3) ? wino_pack_tiles(for_dgrad == 3 ? Cin : Cout, for_dgrad == 3 ? Cout : Cin) : 0;
    const int C = for_dgrad ? Cout : Cin, M = for_dgrad ? Cin : Cout, R = KH * KW;
    if ((C % BK) == 0 && (R == 9 || R == 1)) return (long long)(conv_packed_mpad(M) / 64) * (C / BK);
    const long long total = (long long)((C % BK) == 0 ? conv_packed_mpad(M) : M) * C * R;
    return (total + PACK_TILE_ELEMS - 1) / PACK_TILE_ELEMS;
}

// ----------------------------- host-side dispatch --------------------------
// Launch plan = (N-tile width, split-K count), chosen with a small occupancy model instead of fixed thresholds.
// At these sizes a launch has only a few workgroups per CU, so WAVE QUANTISATION decides the speed: the direct-to-LDS
// kernels keep 64 KB (128x128), 48 KB (128x64 / 64x128) or 32 KB (64x64) of LDS per workgroup = 2 / 3 / 4 resident per CU, a
// CU needs >= 2 resident workgroups (2 waves per SIMD) to keep its MFMA pipe fed across the per-slab barrier, and a CU
// that is left with ONE workgroup runs it at about half speed.  Measured: the conv3_2 weight gradient at 756 workgroups
// (2.95 per CU -> a round of two, then a round of one) ran at 81 TFLOP/s, at 504 (one round of two) 101 TFLOP/s.
//   time(plan) = flops / (model efficiency x 110 TFLOP/s) + split-K slab traffic (write + read) at 3 TB/s
struct LaunchPlan { int bn, splits, bm; };

// tile shape / split count / kernel family of this thread's most recent GEMM-class launch (scda_debug_last_plan: the parity
// tests force every instantiation through SCDA_PLAN_FORCE and must be able to tell that the forced one really ran)
static thread_local int g_last_plan[4] = {0, 0, 0, 0};
static void note_plan(int bm, int bn, int splits, int glds) { g_last_plan[0] = bm; g_last_plan[1] = bn; g_last_plan[2] = splits; g_last_plan[3] = glds; }

static int resident_per_cu(int bm, int bn) { return bm == 256 ? 1 : ((bm == 128 && bn == 128) || bn == 256) ? 2 : (bm == 64 && bn == 64) ? 4 : 3; }
static double tile_efficiency(int bm, int bn) { return bm == 256 ? 1.05 : ((bm == 128 && bn == 128) || bn == 256) ? 1.0 : (bm == 64 && bn == 64) ? 0.85 : 0.90; }
// 1x1 convolutions (one filter tap): a K-slab never re-reads lines the previous slabs brought into L1 / L2 (a 3x3 layer's nine taps
// of a channel group are consecutive slabs over the same input lines), every gather is a fresh L2 / HBM access, and what hides that
// latency is the number of independent workgroups per CU, not the MFMA work per barrier: brute force over the ResNet-50 shapes
// (scripts/tune_plans.py resnet) has the 64x64 tile (4 resident per CU) ahead of the 8-wave 256x128 tile by 7 - 23 % on EVERY 1x1
// forward / data-gradient layer (head 512 -> 2048 on 25088 pixels: 470 vs 561 us), and level with it on every 3x3 layer.
static double tile_efficiency_one_tap(int bm, int bn) { return (bm == 64 && bn == 64) ? 1.0 : (bm == 64 || (bm == 128 && bn == 64)) ? 0.9 : 0.85; }

// time, in units of one workgroup running at full CU speed, for the busiest CU to finish c workgroups with p resident
static double cu_rounds(int c, int p, bool eight_waves = false) {
    if (eight_waves) return (double)c;   // an 8-wave workgroup keeps the CU's MFMA pipes fed on its own
    double t = 0;
    while (c > 0) {
        const int r = c < p ? c : p;
        t += r == 1 ? 2.0 : (double)r;
        c -= r;
    }
    return t;
}

static double plan_cost(long long tiles, int splits, int bm, int bn, double flops, double out_bytes, bool one_tap = false) {
    const long long wgs = tiles * splits;
    const double ideal = (double)wgs / 256.0;
    const double eff = ideal / cu_rounds(cdiv(wgs, 256), resident_per_cu(bm, bn), bm == 256) * (one_tap ? tile_efficiency_one_tap(bm, bn) : tile_efficiency(bm, bn));
    double t = flops / (eff * 110e12);
    if (splits > 1) t += 2.0 * splits * out_bytes / 3e12;
    return t;
}

// bn_lo..bn_hi: candidate N-tile widths (64 and/or 128); must_split: the kernel always writes slabs (weight gradient)
static LaunchPlan plan_search(int M, int N, int K, int bm, bool allow64, bool allow128, bool must_split, size_t ws_bytes,
                              int k_granule, bool allow256, bool allow_bm256, bool one_tap);

// memoised per thread (the same ~60 shapes recur every iteration; launches come from the main and the autograd thread)
static LaunchPlan plan_launch(int M, int N, int K, int bm, bool allow64, bool allow128, bool must_split, size_t ws_bytes,
                              int k_granule, bool allow256 = false, bool allow_bm256 = false, bool one_tap = false) {
    if (const char *f = getenv("SCDA_PLAN_FORCE")) {   // tuning aid: "bm,bn,splits" for every launch (scripts/tune_plans.py)
        int fb = 0, fn = 0, fs = 0;
        if (sscanf(f, "%d,%d,%d", &fb, &fn, &fs) == 3) {
            const bool ok = (fb == bm || (fb == 256 && allow_bm256) || (fb == 64 && getenv("SCDA_PLAN_ALLOW_BM64"))) && ((fn == 64 && allow64) || (fn == 128 && allow128) || (fn == 256 && allow256)) &&
                            !(fb == 256 && fn != 128) && fs >= 1 && (fs == 1 || (size_t)fs * M * N * sizeof(float) <= ws_bytes) &&
                            fs <= (K / (k_granule * 2) > 0 ? K / (k_granule * 2) : 1);
            if (ok) return LaunchPlan{fn, fs, fb};
        }
    }
    if (const char *ov = getenv("SCDA_PLAN_OVERRIDE")) {   // tuning aid: "M,N,K:bm,bn,splits;..." for individual shapes, inside the real iteration
        for (const char *q = ov; q && *q;) {
            int m = 0, n = 0, k = 0, fb = 0, fn = 0, fs = 0;
            if (sscanf(q, "%d,%d,%d:%d,%d,%d", &m, &n, &k, &fb, &fn, &fs) == 6 && m == M && n == N && k == K && !must_split) {
                const bool ok = (fb == bm || (fb == 256 && allow_bm256) || fb == 64) && ((fn == 64 && allow64) || (fn == 128 && allow128) || (fn == 256 && allow256)) &&
                                !(fb == 256 && fn != 128) && fs >= 1 && (fs == 1 || (size_t)fs * M * N * sizeof(float) <= ws_bytes);
                if (ok) return LaunchPlan{fn, fs, fb};
            }
            q = strchr(q, ';');
            if (q) ++q;
        }
    }
    struct Key { int M, N, K, flags; size_t ws; };
    struct Entry { Key k; LaunchPlan p; };
    static thread_local std::vector<Entry> cache;
    const Key key{M, N, K, bm | (allow64 << 8) | (allow128 << 9) | (must_split << 10) | (allow256 << 11) | (k_granule << 12) | (allow_bm256 << 20) | (one_tap << 21), ws_bytes};
    for (const Entry &e : cache)
        if (e.k.M == key.M && e.k.N == key.N && e.k.K == key.K && e.k.flags == key.flags && e.k.ws == key.ws) return e.p;
    const LaunchPlan p = plan_search(M, N, K, bm, allow64, allow128, must_split, ws_bytes, k_granule, allow256, allow_bm256, one_tap);
    if (getenv("SCDA_PLAN_LOG"))
        fprintf(stderr, "[scda plan] M=%d N=%d K=%d %s-> tile %dx%d splits %d (%lld workgroups)\n", M, N, K, must_split ? "wgrad " : "",
                p.bm, p.bn, p.splits, (long long)cdiv(M, p.bm) * cdiv(N, p.bn) * p.splits);
    if (cache.size() < 512) cache.push_back(Entry{key, p});
    return p;
}

static LaunchPlan plan_search(int M, int N, int K, int bm, bool allow64, bool allow128, bool must_split, size_t ws_bytes,
                              int k_granule, bool allow256, bool allow_bm256, bool one_tap) {
    const double flops = 2.0 * M * (double)N * K, out_bytes = (double)M * N * sizeof(float);
    double best_t = 1e30;
    int max_s = K / (k_granule * 4);   // at least 4 K-steps per split
    if (max_s < 1) max_s = 1;
    if (max_s > 256) max_s = 256;
    while (max_s > 1 && (size_t)max_s * M * N * sizeof(float) > ws_bytes) --max_s;
    LaunchPlan best{allow128 ? 128 : 64, 1, bm};
    // tile-row candidates: the natural one; 256 (8 waves) where legal; 64 for 65..128-row problems (the decoder's 128-channel
    // layers: 512 half-height tiles and no split-K beat 128 full tiles split four ways by ~6 %)
    // ... and for 1x1 convolutions of any height (tile_efficiency_one_tap)
    const bool allow_bm64 = bm == 128 && (M <= 128 || one_tap) && allow64 && !must_split && k_granule == BK;
    for (int pass = 0; pass < 3; ++pass) {
        if ((pass == 1 && !allow_bm256) || (pass == 2 && !allow_bm64)) continue;
        const int tbm = pass == 1 ? 256 : pass == 2 ? 64 : bm;
        for (int bn = 64; bn <= 256; bn *= 2) {
            if ((bn == 64 && !allow64) || (bn == 128 && !allow128) || (bn == 256 && !allow256)) continue;
            if (tbm == 256 && bn != 128) continue;
            const long long tiles = (long long)cdiv(M, tbm) * cdiv(N, bn);
            for (int sp = 1; sp <= max_s; ++sp) {
                if (tiles * sp > 4096 && sp > 1) break;   // plenty of workgroups already: splitting only adds traffic
                double t = plan_cost(tiles, sp, tbm, bn, flops, out_bytes, one_tap);
                if (must_split && sp == 1) t += 2.0 * out_bytes / 3e12;
                if (t < best_t) { best_t = t; best = LaunchPlan{bn, sp, tbm}; }
            }
        }
    }
    return best;
}

static int round_k_per_split(int K, int splits) {
    int kps = (K + splits - 1) / splits;
    kps = (kps + BK - 1) / BK * BK;
    return kps;
}

template <int KH, int KW, int S, bool DGRAD>
static int launch_conv(const float *Wm, const float *X, const ConvGeom &g0, Epi e, float *ws, size_t ws_bytes,
                       hipStream_t st) {
    ConvGeom g = g0;
    // stride-2 data gradient on the direct-to-LDS kernel: class-major N axis, a quarter of the K-slabs per tile (ConvGeom::parity)
    static const bool no_parity = getenv("SCDA_CONV_NO_PARITY") != nullptr;   // A/B knob
    const bool parity = S == 2 && DGRAD && g.slab_aligned && (g.PH % 2) == 0 && (g.PW % 2) == 0 && KH * KW <= 32 && !no_parity;
    g.parity = 0; g.nc = g.ncp = 0;
    if (g.slab_aligned) {   // the LDS-DMA kernel addresses the images one tile touches with 32-bit lane offsets
        const long long phw = (long long)g.PH * g.PW / (parity ? 4 : 1), per_image = (long long)g.CB * g.HB * g.WB * 4;
        const long long images = std::min<long long>(g.batch, (256 + phw - 1) / phw + 1);
        if (images * per_image >= (1LL << 31)) {
            set_error("conv: %lld x %lld bytes of gathered tensor under one tile exceed the 2 GB a buffer descriptor addresses", images, per_image);
            return SCDA_EINVAL;
        }
    }
    const bool small_m = g.M <= 64;
    const int BMv = small_m ? 64 : 128;
    static const char *force = getenv("SCDA_CONV_BN");   // experiment knob: 64 | 128
    const int fbn = force ? (atoi(force) == 64 ? 64 : 128) : 0;
    const char *fbm = getenv("SCDA_CONV_BM");            // experiment / test knob: 256 forces the 8-wave tile where legal
    const bool bm256_ok = g.slab_aligned && (g.M % 256) == 0 && !fbn;
    static const bool no_one_tap = getenv("SCDA_PLAN_NO_ONE_TAP") != nullptr;   // A/B knob
    // parity classes: a launch is four GEMMs of N / 4 pixels and (on average) K / 4 each
    const int planN = parity ? g.N / 4 : g.N, planK = parity ? std::max(BK, g.K / 4 / BK * BK) : g.K;
    LaunchPlan plan = plan_launch(g.M, planN, planK, BMv, fbn != 128, fbn != 64, false, ws_bytes / (parity ? 4 : 1), BK,
                                  small_m && g.slab_aligned && !fbn, bm256_ok && !(fbm && atoi(fbm) != 256),
                                  KH * KW == 1 && g.slab_aligned && !no_one_tap);
    if (bm256_ok && fbm && atoi(fbm) == 256 && plan.bm != 256) {
        plan.bm = 256; plan.bn = 128;
        while (plan.splits > 1 && (size_t)plan.splits * g.M * g.N * sizeof(float) > ws_bytes) --plan.splits;
    }
    // <= 32 output rows on many pixels (the decoders' 64 -> 32 stage: 262144 pixels; the data gradient into the discriminators'
    // 32-channel map): the 32 x 256 tile, one K pass.  SCDA_PLAN_FORCE=32,256,1 forces it wherever it is legal (tests),
    // any other forced plan keeps it out.
    {
        static const bool no_bm32 = getenv("SCDA_CONV_NO_BM32") != nullptr;   // A/B knob
        const char *f = getenv("SCDA_PLAN_FORCE");
        int fb = 0, fnn = 0, fs = 0;
        const bool forced = f && sscanf(f, "%d,%d,%d", &fb, &fnn, &fs) == 3;
        const bool legal = g.M <= 32 && g.slab_aligned && !fbn;
        if (legal && (forced ? (fb == 32 && fnn == 256 && fs == 1) : (!no_bm32 && planN >= (parity ? 64 * 256 : 256 * 256)))) plan = LaunchPlan{256, 1, 32};
    }
    const int BNv = plan.bn;
    const int BMt = plan.bm;                             // tile rows of this launch (BMv unless the 8-wave tile was chosen)
    int splits = plan.splits;
    g.k_per_split = round_k_per_split(g.K, splits);
    splits = cdiv(g.K, g.k_per_split);
    g.nx = cdiv(g.N, BNv); g.ny = cdiv(g.M, BMt); g.swz = xcd_swizzle_enabled();
    // many M-tiles whose weight panels together do not fit an XCD's L2 (the ResNet RoI head's 512 -> 2048 1x1: 32 panels = 4.2 MB,
    // walked M-tile-fastest they were re-fetched for every pixel tile: 762 MB read for a 51 MB input): the grouped order of the
    // dense GEMMs (tile_coords: 8 M-tiles at a time across all pixel tiles -- the input is read once per group instead)
    static const bool no_group = getenv("SCDA_CONV_NO_MGROUP") != nullptr;   // A/B knob
    if (!no_group && g.swz && !parity && g.ny >= 16 && (long long)g.nx * g.ny >= 1024 && (double)g.M * g.K * sizeof(float) > 4e6)
        g.swz |= 2;
    if (parity) {
        const int pq = (g.PH / 2) * (g.PW / 2);
        g.parity = 1;
        g.nc = g.batch * pq;
        g.ncp = cdiv(g.nc, BNv) * BNv;
        g.dNCP = Div(g.ncp); g.dPQ = Div(pq); g.dPQW = Div(g.PW / 2);
        g.nx = 4 * (g.ncp / BNv);
        splits = std::max(1, std::min(plan.splits, (g.CB / BK) * KH * KW));   // the kernel divides each class's slabs evenly
        while (splits > 1 && (size_t)splits * g.M * g.N * sizeof(float) > ws_bytes) --splits;
    }
    e.splits = splits;
    e.ws = ws;
    dim3 grid((unsigned)g.nx * g.ny * splits);
    note_plan(BMt, BNv, splits, g.slab_aligned);
    prof_begin(g.slab_aligned ? PK_CONV + ((DGRAD ? 2 : 0) + (BMt <= 64 ? 1 : 0)) * 3 + prof_shape(KH, S) : (int)PK_CONV_GATHER,
               2.0 * g.M * (double)g.N * g.K, st,
               4.0 * ((double)g.batch * g.CB * g.HB * g.WB + (double)g.M * g.K + (double)g.M * g.N));
    g.mpad = conv_packed_mpad(g.M);
#define CONV_LAUNCH(BM_, BN_)                                                                                            \
    do {                                                                                                                 \
        if (g.slab_aligned)                                                                                              \
            hipLaunchKernelGGL((conv_igemm_glds_kernel<BM_, BN_, KH, KW, S, DGRAD>), grid, dim3(ConvGldsCfg<BM_, BN_>::THREADS), 0, st, Wm, X, g, e);  \
        else                                                                                                             \
            hipLaunchKernelGGL((conv_igemm_kernel<BM_, BN_, KH, KW, S, DGRAD>), grid, dim3(256), 0, st, Wm, X, g, e);        \
    } while (0)
    if (BMt == 256) hipLaunchKernelGGL((conv_igemm_glds_kernel<256, 128, KH, KW, S, DGRAD>), grid, dim3(ConvGldsCfg<256, 128>::THREADS), 0, st, Wm, X, g, e);
    else if (BMt == 32) hipLaunchKernelGGL((conv_igemm_glds_kernel<32, 256, KH, KW, S, DGRAD>), grid, dim3(ConvGldsCfg<32, 256>::THREADS), 0, st, Wm, X, g, e);
    else if (BMt == 64 && BNv == 256) hipLaunchKernelGGL((conv_igemm_glds_kernel<64, 256, KH, KW, S, DGRAD>), grid, dim3(ConvGldsCfg<64, 256>::THREADS), 0, st, Wm, X, g, e);
    else if (BMt == 64 && BNv == 64) CONV_LAUNCH(64, 64);
    else if (BMt == 64) CONV_LAUNCH(64, 128);
    else if (BNv == 64) CONV_LAUNCH(128, 64);
    else CONV_LAUNCH(128, 128);
#undef CONV_LAUNCH
    prof_end(st);
    int rc = launch_status("conv_igemm_kernel");
    if (rc || splits == 1) return rc;
    const bool vec = (g.N & 3) == 0 && (g.dPHW.d & 3) == 0 && ((((uintptr_t)ws) | ((uintptr_t)e.out) | ((uintptr_t)e.mask_src)) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(conv_splitk_reduce_kernel<true>, dim3(ew_grid(((long long)g.M * g.N) >> 2)), dim3(256), 0, st, ws, splits, g.M,
                           g.N, g.dPHW, e.bias, e.act, e.slope, e.out, e.mask_src, e.mask_slope);
    else
        hipLaunchKernelGGL(conv_splitk_reduce_kernel<false>, dim3(ew_grid((long long)g.M * g.N)), dim3(256), 0, st, ws, splits, g.M,
                           g.N, g.dPHW, e.bias, e.act, e.slope, e.out, e.mask_src, e.mask_slope);
    return launch_status("conv_splitk_reduce_kernel");
}

// the same combine for split-K slabs written by another translation unit (conv_wino.hip): ws [splits][M][N] in the natural
// pixel order, N = batch * phw
int launch_conv_reduce(const float *ws, int splits, int M, int N, int phw, const float *bias, int act, float slope, float *out,
                       const float *mask_src, float mask_slope, hipStream_t st) {
    const Div d(phw);
    const bool vec = (N & 3) == 0 && (phw & 3) == 0 && ((((uintptr_t)ws) | ((uintptr_t)out) | ((uintptr_t)mask_src)) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(conv_splitk_reduce_kernel<true>, dim3(ew_grid(((long long)M * N) >> 2)), dim3(256), 0, st, ws, splits, M, N, d, bias,
                           act, slope, out, mask_src, mask_slope);
    else
        hipLaunchKernelGGL(conv_splitk_reduce_kernel<false>, dim3(ew_grid((long long)M * N)), dim3(256), 0, st, ws, splits, M, N, d, bias, act,
                           slope, out, mask_src, mask_slope);
    return launch_status("conv_splitk_reduce_kernel");
}

// the weight gradients' combine (slabs [splits][M][N] + fused bias-gradient partials) for conv_wino.hip
int launch_wgrad_reduce(const float *ws, int splits, long long total, int N, int accumulate, float *out, const float *db_ws, float *db,
                        int db_n, int db_accumulate, hipStream_t st) {
    return launch_dense_reduce(ws, splits, total, N, nullptr, 0, (int)ACT_NONE, 0.f, accumulate, out, db_ws, db, db_n, db_accumulate, st);
}

template <int KH, int KW, int S>
static int launch_wgrad(const float *dY, const float *X, WgradGeom g, float *dW, int accumulate, float *ws,
                        size_t ws_bytes, hipStream_t st, float *db = nullptr, int db_accumulate = 0) {
    if (db) {   // room for the bias-gradient partials [splits <= 256][M] behind the slabs
        const size_t need = (size_t)256 * g.M * sizeof(float);
        if (ws_bytes <= 2 * need) { set_error("conv wgrad: workspace too small for the fused bias gradient"); return SCDA_EINVAL; }
        ws_bytes -= need;
    }
    const bool small = g.M <= 64;
    const int BNv = (g.N <= 64) ? 64 : 128;
    static const bool no_glds = getenv("SCDA_WGRAD_NO_GLDS") != nullptr;   // A/B knob
    // the LDS-DMA kernel addresses one image of dY / X through a buffer descriptor with 32-bit lane offsets
    const bool fits_2g = (long long)g.Cout * g.OH * g.OW * 4 < (1LL << 31) && (long long)g.Cin * g.IH * g.IW * 4 < (1LL << 31);
    // K-slabs of 16 pixels must not straddle two images: OH*OW % 16 == 0.  (A partial last slab for batch-1 planes was built and
    // measured: ResNet layer3's 50 x 84 weight gradients took 325 us on this kernel's general addressing path against 162 us on the
    // register-staged one -- those layers have only 4200 pixels of K to amortise the pipeline over; removed.)
    const bool glds = !no_glds && g.a_vec4 && (g.dOHW.d % BK) == 0 && fits_2g;
    const char *fbm = getenv("SCDA_CONV_BM");                              // 256 forces the 8-wave tile where legal
    const bool bm256_ok = glds && BNv == 128 && (g.M % 256) == 0;
    LaunchPlan plan = plan_launch(g.M, g.N, g.K, small ? 64 : 128, BNv == 64, BNv == 128, true, ws_bytes, 32, false,
                                  bm256_ok && !(fbm && atoi(fbm) != 256));
    if (bm256_ok && fbm && atoi(fbm) == 256) plan.bm = 256;
    // <= 32 output channels on the direct-to-LDS kernel: the 32 x 128 tile, same split count (SCDA_PLAN_FORCE=32,128,s forces it where
    // legal; any other forced plan, or SCDA_WGRAD_NO_BM32, keeps it out)
    {
        static const bool no_bm32 = getenv("SCDA_WGRAD_NO_BM32") != nullptr;
        const char *f = getenv("SCDA_PLAN_FORCE");
        int fb = 0, fnn = 0, fs = 0;
        const bool forced = f && sscanf(f, "%d,%d,%d", &fb, &fnn, &fs) == 3;
        const bool legal = glds && g.M <= 32 && BNv == 128;
        if (legal && (forced ? (fb == 32 && fnn == 128) : !no_bm32)) {
            plan.bm = 32;
            if (forced && fs >= 1 && (size_t)fs * g.M * g.N * sizeof(float) <= ws_bytes) plan.splits = fs;
        }
    }
    const int BMv = plan.bm;
    int splits = plan.splits;
    if ((size_t)splits * g.M * g.N * sizeof(float) > ws_bytes) { set_error("conv wgrad: workspace too small"); return SCDA_EINVAL; }
    g.k_per_split = (round_k_per_split(g.K, splits) + 31) / 32 * 32;
    splits = cdiv(g.K, g.k_per_split);
    g.nx = cdiv(g.N, BNv); g.ny = cdiv(g.M, BMv); g.swz = xcd_swizzle_enabled();
    dim3 grid((unsigned)g.nx * g.ny * splits);
    note_plan(BMv, BNv, splits, glds);
    prof_begin(PK_CONV_WGRAD + prof_shape(KH, S), 2.0 * g.M * (double)g.N * g.K, st);
    static const char *wbk_env = getenv("SCDA_WGRAD_BK");
    // measured (SCDA_WGRAD_BK=16|32 A/B): 32-deep slabs gain 10-17 % for the 64-row tiles (conv1_x, decoder heads), lose
    // up to 8 % for 128-row tiles
    const bool bk32 = (wbk_env ? atoi(wbk_env) == 32 : small) && (g.k_per_split % 32) == 0;
#define WGRAD_LAUNCH(BM_, BN_)                                                                                           \
    do {                                                                                                                 \
        if (bk32) hipLaunchKernelGGL((conv_wgrad_kernel<BM_, BN_, KH, KW, S, 32>), grid, dim3(256), 0, st, dY, X, g, ws);   \
        else hipLaunchKernelGGL((conv_wgrad_kernel<BM_, BN_, KH, KW, S, 16>), grid, dim3(256), 0, st, dY, X, g, ws);        \
    } while (0)
    if (db && !glds) { set_error("conv wgrad: the fused bias gradient needs OH*OW %% 16 == 0 and 16-byte aligned dy"); return SCDA_EINVAL; }
    float *db_ws = db ? ws + (size_t)splits * g.M * g.N : nullptr;
#define WGRAD_GLDS_LAUNCH(BM_, BN_) hipLaunchKernelGGL((conv_wgrad_glds_kernel<BM_, BN_, KH, KW, S>), grid, dim3(WgradGldsCfg<BM_, BN_>::THREADS), 0, st, dY, X, g, ws, db_ws)
    if (glds && BMv == 256) {
        hipLaunchKernelGGL((conv_wgrad_glds_kernel<256, 128, KH, KW, S>), grid, dim3(WgradGldsCfg<256, 128>::THREADS), 0, st, dY, X, g, ws, db_ws);
    } else if (glds && BMv == 32) {
        WGRAD_GLDS_LAUNCH(32, 128);
    } else if (glds) {
        if (small && BNv == 64) WGRAD_GLDS_LAUNCH(64, 64);
        else if (small) WGRAD_GLDS_LAUNCH(64, 128);
        else if (BNv == 64) WGRAD_GLDS_LAUNCH(128, 64);
        else WGRAD_GLDS_LAUNCH(128, 128);
    } else if (small && BNv == 64) WGRAD_LAUNCH(64, 64);
    else if (small) WGRAD_LAUNCH(64, 128);
    else if (BNv == 64) WGRAD_LAUNCH(128, 64);
    else WGRAD_LAUNCH(128, 128);
#undef WGRAD_LAUNCH
#undef WGRAD_GLDS_LAUNCH
    prof_end(st);
    int rc = launch_status("conv_wgrad_kernel");
    if (rc) return rc;
    const long long total = (long long)g.M * g.N;
    return launch_dense_reduce(ws, splits, total, g.N, nullptr, 0, (int)ACT_NONE, 0.f, accumulate, dW, db_ws, db, g.M,
                               db_accumulate, st);
}

}  // namespace scda

using namespace scda;

#define CONV_DISPATCH(FN, ...)                                                                   \
    if (KH == 3 && KW == 3 && S == 1) return FN<3, 3, 1 __VA_ARGS__;                              \
    if (KH == 3 && KW == 3 && S == 2) return FN<3, 3, 2 __VA_ARGS__;                              \
    if (KH == 1 && KW == 1 && S == 1) return FN<1, 1, 1 __VA_ARGS__;                              \
    if (KH == 1 && KW == 1 && S == 2) return FN<1, 1, 2 __VA_ARGS__;   /* ResNet down-sampling shortcuts */ \
    set_error("conv: unsupported kernel %dx%d stride %d (supported: 3x3 s1, 3x3 s2, 1x1 s1, 1x1 s2; forward also 7x7 s2)", KH, KW, S); \
    return SCDA_EINVAL;

static int conv_out_dim(int in, int k, int s, int p) { return (in + 2 * p - k) / s + 1; }

// row_period argument of the conv entry points: the image is a stack of independent maps of `period` rows (ConvGeom::row_period).
// validated period (0: plain image); -1 = illegal for this convolution
static int take_row_period(const char *who, int period, int IH, int KH, int KW, int S, int P) {
    if (period <= 0) return 0;
    if (S != 1 || KH != KW || 2 * P != KH - 1 || (IH % period) != 0) {
        set_error("%s: row period %d needs a stride-1 same-size convolution on an image whose height (%d) is a multiple of it", who, period, IH);
        return -1;
    }
    return KH == 1 ? 0 : period;   // a 1x1 convolution has no neighbours to protect
}

SCDA_API size_t scda_conv2d_workspace_bytes(int batch, int Cin, int IH, int IW, int Cout, int KH, int KW, int S, int P) {
    // enough for: fwd/dgrad split-K slabs and wgrad slabs (<= 64 splits of the weight matrix,
    // bounded by 8 output-sized slabs)
    const int OH = conv_out_dim(IH, KH, S, P), OW = conv_out_dim(IW, KW, S, P);
    size_t out_elems = (size_t)batch * Cout * OH * OW, in_elems = (size_t)batch * Cin * IH * IW;
    size_t w_elems = (size_t)Cout * Cin * KH * KW;
    size_t a = 8 * (out_elems > in_elems ? out_elems : in_elems);
    size_t b = (w_elems <= 128 * 128 * 9 ? 256 : 64) * w_elems;
    size_t cap = (size_t)256 << 20;  // slabs never need to exceed 256 MB: pick_splits shrinks to fit
    size_t need = (a > b ? a : b) * sizeof(float);
    if (need > cap) need = cap;
    size_t minimum = 2 * w_elems * sizeof(float);
    return need > minimum ? need : minimum;
}

static int conv1x1_as_x9(const float *A, const float *B, float *C, int M, int N, int K, int lda, int ldb, int trans_a, int trans_b,
                         const float *bias, int act, float slope, int accumulate, void *ws, size_t ws_bytes, hipStream_t st);

SCDA_API int scda_conv2d_fwd_hip(const float *x, const float *w, const float *bias, float *y, int batch, int Cin, int IH,
                                 int IW, int Cout, int KH, int KW, int S, int P, int row_period_arg, int act, float slope, void *ws,
                                 size_t ws_bytes, void *stream) {
    const int row_period = take_row_period("scda_conv2d_fwd_hip", row_period_arg, IH, KH, KW, S, P);
    if (row_period < 0) return SCDA_EINVAL;
    if (!x || !w || !y || batch <= 0 || Cin <= 0 || Cout <= 0) { set_error("scda_conv2d_fwd_hip: bad arguments"); return SCDA_EINVAL; }
    if (!zero_page()) { set_error("scda_conv2d_fwd_hip: could not allocate the zero page"); return SCDA_ELAUNCH; }
    const int OH = conv_out_dim(IH, KH, S, P), OW = conv_out_dim(IW, KW, S, P);
    if (OH <= 0 || OW <= 0) { set_error("scda_conv2d_fwd_hip: empty output"); return SCDA_EINVAL; }
    ConvGeom g;
    g.batch = batch; g.CB = Cin; g.HB = IH; g.WB = IW; g.PH = OH; g.PW = OW; g.pad = P;
    g.M = Cout; g.N = batch * OH * OW; g.K = Cin * KH * KW; g.k_per_split = 0;
    g.dPHW = Div(OH * OW); g.dPW = Div(OW); g.dCB = Div(Cin);
    g.slab_aligned = (Cin % BK) == 0;
    g.zp = zero_page();
    g.row_period = row_period;
    if (batch == 1 && KH == 1 && KW == 1 && S == 1 && P == 0 && g.slab_aligned) {      // w is packed [Cin][mpad]: the GEMM's A stored [K][M]
        const int rc = conv1x1_as_x9(w, x, y, Cout, IH * IW, Cin, conv_packed_mpad(Cout), IH * IW, 1, 1, bias, act, slope, 0, ws, ws_bytes, as_stream(stream));
        if (rc <= 0) return rc;
    }
    static const bool no_direct = getenv("SCDA_CONV_NO_SMALL_CIN_FWD") != nullptr;      // A/B knob
    if (!no_direct && KH == 3 && KW == 3 && S == 1 && P == 1 && Cin <= 4 && !g.slab_aligned && row_period == 0 && IW >= 64) {
        // image-side 3x3 layer (VGG conv1_1): direct kernel on the tap-major [Cout][9 Cin] weights the gather kernel takes
        const dim3 grid((unsigned)cdiv(IW, 256), (unsigned)IH, (unsigned)batch);
        hipStream_t st = as_stream(stream);
        note_plan(0, 0, 1, 0);
        prof_begin(PK_CONV_GATHER, 2.0 * Cout * (double)g.N * g.K, st);
        if (Cin == 1) hipLaunchKernelGGL((conv3x3_small_cin_fwd_kernel<1>), grid, dim3(256), 0, st, x, w, bias, y, IH, IW, Cout, act, slope);
        else if (Cin == 2) hipLaunchKernelGGL((conv3x3_small_cin_fwd_kernel<2>), grid, dim3(256), 0, st, x, w, bias, y, IH, IW, Cout, act, slope);
        else if (Cin == 3) hipLaunchKernelGGL((conv3x3_small_cin_fwd_kernel<3>), grid, dim3(256), 0, st, x, w, bias, y, IH, IW, Cout, act, slope);
        else hipLaunchKernelGGL((conv3x3_small_cin_fwd_kernel<4>), grid, dim3(256), 0, st, x, w, bias, y, IH, IW, Cout, act, slope);
        prof_end(st);
        return launch_status("conv3x3_small_cin_fwd_kernel");
    }
    Epi e{y, nullptr, bias, 0, act, slope, 1, 0, nullptr, 0.f};
    if (KH == 7 && KW == 7 && S == 2)   // the ResNet stem (3 -> 64, frozen in the reference: models/mask_rcnn/resnet.py:230-238): forward only
        return launch_conv<7, 7, 2, false>(w, x, g, e, (float *)ws, ws_bytes, as_stream(stream));
    CONV_DISPATCH(launch_conv, , false > (w, x, g, e, (float *)ws, ws_bytes, as_stream(stream)))
}

// dx = dgrad(dy, wt) where wt = pack(w, for_dgrad=1) is [Cin][KH*KW][Cout] (scda_conv2d_pack_weight_hip)
SCDA_API int scda_conv2d_dgrad_hip(const float *dy, const float *wt, float *dx, int batch, int Cin, int IH, int IW,
                                   int Cout, int KH, int KW, int S, int P, int row_period, void *ws, size_t ws_bytes, void *stream) {
    return scda_conv2d_dgrad_act_hip(dy, wt, dx, batch, Cin, IH, IW, Cout, KH, KW, S, P, row_period, nullptr, 0.f, ws, ws_bytes, stream);
}

// ... with the activation gradient of the layer that PRODUCED this conv's input folded into the epilogue:
// dx = dgrad(dy) * (act_src > 0 ? 1 : act_slope), act_src = the conv's input x [batch,Cin,IH,IW] (a ReLU / LeakyReLU output)
SCDA_API int scda_conv2d_dgrad_act_hip(const float *dy, const float *wt, float *dx, int batch, int Cin, int IH, int IW,
                                       int Cout, int KH, int KW, int S, int P, int row_period_arg, const float *act_src, float act_slope,
                                       void *ws, size_t ws_bytes, void *stream) {
    const int row_period = take_row_period("scda_conv2d_dgrad_hip", row_period_arg, IH, KH, KW, S, P);
    if (row_period < 0) return SCDA_EINVAL;
    if (!dy || !wt || !dx || batch <= 0) { set_error("scda_conv2d_dgrad_hip: bad arguments"); return SCDA_EINVAL; }
    const int OH = conv_out_dim(IH, KH, S, P), OW = conv_out_dim(IW, KW, S, P);
    ConvGeom g;
    g.batch = batch; g.CB = Cout; g.HB = OH; g.WB = OW; g.PH = IH; g.PW = IW; g.pad = P;
    g.M = Cin; g.N = batch * IH * IW; g.K = Cout * KH * KW; g.k_per_split = 0;
    g.dPHW = Div(IH * IW); g.dPW = Div(IW); g.dCB = Div(Cout);
    g.slab_aligned = (Cout % BK) == 0;
    g.zp = zero_page();
    g.row_period = row_period;
    if (!g.zp) { set_error("scda_conv2d_dgrad_hip: could not allocate the zero page"); return SCDA_ELAUNCH; }
    if (batch == 1 && KH == 1 && KW == 1 && S == 1 && P == 0 && g.slab_aligned && !act_src) {   // wt is packed [Cout][mpad(Cin)]
        const int rc = conv1x1_as_x9(wt, dy, dx, Cin, IH * IW, Cout, conv_packed_mpad(Cin), IH * IW, 1, 1, nullptr, (int)ACT_NONE, 0.f, 0, ws, ws_bytes, as_stream(stream));
        if (rc <= 0) return rc;
    }
    Epi e{dx, nullptr, nullptr, 0, (int)ACT_NONE, 0.f, 1, 0, act_src, act_slope};
    CONV_DISPATCH(launch_conv, , true > (wt, dy, g, e, (float *)ws, ws_bytes, as_stream(stream)))
}

// dx [batch,Cin<=4,IH,IW] = data gradient of dy [batch,Cout,OH,OW]; w is the UNPACKED [Cout,Cin,KH,KW] weight
SCDA_API int scda_conv2d_dgrad_small_cin_hip(const float *dy, const float *w, float *dx, int batch, int Cin, int IH, int IW,
                                             int Cout, int KH, int KW, int S, int P, void *stream) {
    if (!dy || !w || !dx || batch <= 0 || Cin <= 0 || Cin > 4 || Cout <= 0 || S <= 0) { set_error("scda_conv2d_dgrad_small_cin_hip: bad arguments"); return SCDA_EINVAL; }
    const int OH = conv_out_dim(IH, KH, S, P), OW = conv_out_dim(IW, KW, S, P);
    const size_t lds = (size_t)Cout * KH * KW * 4 * sizeof(float);
    if (lds > 64 * 1024) { set_error("scda_conv2d_dgrad_small_cin_hip: Cout=%d too large", Cout); return SCDA_EINVAL; }
    const long long total = (long long)batch * IH * IW;
    hipStream_t st = as_stream(stream);
    if (KH == 3 && KW == 3)
        hipLaunchKernelGGL((conv_dgrad_small_cin_kernel<3, 3>), dim3(ew_grid(total) * 4), dim3(256), lds, st, dy, w, dx, batch, Cin, IH, IW, Cout, OH, OW, S, P);
    else if (KH == 1 && KW == 1)
        hipLaunchKernelGGL((conv_dgrad_small_cin_kernel<1, 1>), dim3(ew_grid(total) * 4), dim3(256), lds, st, dy, w, dx, batch, Cin, IH, IW, Cout, OH, OW, S, P);
    else { set_error("scda_conv2d_dgrad_small_cin_hip: unsupported kernel %dx%d", KH, KW); return SCDA_EINVAL; }
    return launch_status("conv_dgrad_small_cin_kernel");
}

SCDA_API size_t scda_conv2d_packed_elems(int Cout, int Cin, int KH, int KW, int for_dgrad) {
    if (for_dgrad >= 2) return (KH == 3 && KW == 3) ? (size_t)wino_packed_elems(for_dgrad == 3 ? Cin : Cout, for_dgrad == 3 ? Cout : Cin) : 0;
    const int C = for_dgrad ? Cout : Cin, M = for_dgrad ? Cin : Cout;
    const int mpad = (C % BK) == 0 ? conv_packed_mpad(M) : M;
    return (size_t)mpad * C * KH * KW;
}

SCDA_API int scda_conv2d_pack_weight_hip(const float *w, float *out, int Cout, int Cin, int KH, int KW, int for_dgrad,
                                         void *stream) {
    if (!w || !out || Cout <= 0 || Cin <= 0) { set_error("scda_conv2d_pack_weight_hip: bad arguments"); return SCDA_EINVAL; }
    const long long total = (long long)scda_conv2d_packed_elems(Cout, Cin, KH, KW, for_dgrad);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(ew_grid(total)), dim3(256), 0, as_stream(stream), w, out, Cout, Cin, KH * KW,
                       for_dgrad, total);
    return launch_status("pack_weight_kernel");
}

SCDA_API int scda_conv2d_pack_weights_batched_hip(const float *base, float *out, const long long *desc, int n,
                                                  long long n_tiles, void *stream) {
    if (!base || !out || !desc || n <= 0 || n_tiles <= 0 || n_tiles > 0x7fffffffLL) { set_error("scda_conv2d_pack_weights_batched_hip: bad arguments"); return SCDA_EINVAL; }
    hipLaunchKernelGGL(pack_weights_batched_kernel, dim3((unsigned)n_tiles), dim3(256), 0, as_stream(stream), base, out, desc, n);
    return launch_status("pack_weights_batched_kernel");
}

SCDA_API int scda_conv2d_wgrad_hip(const float *dy, const float *x, float *dw, int batch, int Cin, int IH, int IW,
                                   int Cout, int KH, int KW, int S, int P, int row_period_arg, int accumulate, void *ws, size_t ws_bytes,
                                   void *stream) {
    const int row_period = take_row_period("scda_conv2d_wgrad_hip", row_period_arg, IH, KH, KW, S, P);
    if (row_period < 0) return SCDA_EINVAL;
    if (!dy || !x || !dw || !ws) { set_error("scda_conv2d_wgrad_hip: bad arguments"); return SCDA_EINVAL; }
    const int OH = conv_out_dim(IH, KH, S, P), OW = conv_out_dim(IW, KW, S, P);
    WgradGeom g;
    g.batch = batch; g.Cin = Cin; g.IH = IH; g.IW = IW; g.Cout = Cout; g.OH = OH; g.OW = OW; g.pad = P;
    g.M = Cout; g.N = Cin * KH * KW; g.K = batch * OH * OW; g.k_per_split = 0;
    g.dOHW = Div(OH * OW); g.dOW = Div(OW);
    g.a_vec4 = ((OH * OW) % 4) == 0 && (((uintptr_t)dy) & 15) == 0;
    g.zp = zero_page();
    g.row_period = row_period;
    if (!g.zp) { set_error("scda_conv2d_wgrad_hip: could not allocate the zero page"); return SCDA_ELAUNCH; }
    if (batch == 1 && KH == 1 && KW == 1 && S == 1 && P == 0) {      // dW[Cout][Cin] (+)= dY[Cout][HW] X[Cin][HW]^T: both K-contiguous
        const int rc = conv1x1_as_x9(dy, x, dw, Cout, Cin, IH * IW, IH * IW, IH * IW, 0, 0, nullptr, (int)ACT_NONE, 0.f, accumulate, ws, ws_bytes, as_stream(stream));
        if (rc <= 0) return rc;
    }
    CONV_DISPATCH(launch_wgrad, > (dy, x, g, dw, accumulate, (float *)ws, ws_bytes, as_stream(stream)))
}

SCDA_API int scda_conv2d_wgrad_bias_fusable(int batch, int Cout, int OH, int OW, const float *dy) {
    (void)batch; (void)Cout;
    return ((OH * OW) % BK) == 0 && (((uintptr_t)dy) & 15) == 0 && !getenv("SCDA_WGRAD_NO_GLDS") && !getenv("SCDA_WGRAD_NO_BIAS_FUSE");
}

SCDA_API int scda_conv2d_wgrad_bias_hip(const float *dy, const float *x, float *dw, float *db, int batch, int Cin, int IH,
                                        int IW, int Cout, int KH, int KW, int S, int P, int row_period_arg, int accumulate, int db_accumulate,
                                        void *ws, size_t ws_bytes, void *stream) {
    const int row_period = take_row_period("scda_conv2d_wgrad_bias_hip", row_period_arg, IH, KH, KW, S, P);
    if (row_period < 0) return SCDA_EINVAL;
    if (!dy || !x || !dw || !db || !ws) { set_error("scda_conv2d_wgrad_bias_hip: bad arguments"); return SCDA_EINVAL; }
    const int OH = conv_out_dim(IH, KH, S, P), OW = conv_out_dim(IW, KW, S, P);
    WgradGeom g;
    g.batch = batch; g.Cin = Cin; g.IH = IH; g.IW = IW; g.Cout = Cout; g.OH = OH; g.OW = OW; g.pad = P;
    g.M = Cout; g.N = Cin * KH * KW; g.K = batch * OH * OW; g.k_per_split = 0;
    g.dOHW = Div(OH * OW); g.dOW = Div(OW);
    g.a_vec4 = ((OH * OW) % 4) == 0 && (((uintptr_t)dy) & 15) == 0;
    g.zp = zero_page();
    g.row_period = row_period;
    if (!g.zp) { set_error("scda_conv2d_wgrad_bias_hip: could not allocate the zero page"); return SCDA_ELAUNCH; }
    CONV_DISPATCH(launch_wgrad, > (dy, x, g, dw, accumulate, (float *)ws, ws_bytes, as_stream(stream), db, db_accumulate))
}

SCDA_API void scda_debug_last_plan(int *out4) {
    for (int i = 0; i < 4; ++i) out4[i] = g_last_plan[i];
}

static int x9_persistent_workgroups() {
    static const int n_cu = [] { int d = 0, n = 0; if (hipGetDevice(&d) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n < 8) n = 256; return n / 8 * 8; }();
    return n_cu;
}

SCDA_API size_t scda_gemm_workspace_bytes(int M, int N, int K) {
    (void)K;
    // split-K slabs, or the stream-K form's two partial-tile slots per persistent workgroup (gemm_x9_kernel)
    return std::max((size_t)16 * M * N * sizeof(float), (size_t)2 * x9_persistent_workgroups() * GemmX9Cfg::BM * GemmX9Cfg::BN * sizeof(float));
}

static int gemm_x9_launch(const float *A, const float *B, float *C, int M, int N, int K, int lda, int ldb, int ldc, int trans_a, int trans_b,
                          const float *bias, int bias_on_n, int act, float slope, int accumulate, void *ws, size_t ws_bytes, hipStream_t st) {
    using X = GemmX9Cfg;
    const int nx = cdiv(N, X::BN), ny = cdiv(M, X::BM);
    const long long tiles = (long long)nx * ny;
    const int n_cu = x9_persistent_workgroups();
    // more tiles than CUs: the persistent stream-K form (SCDA_GEMM_X9_SK=0 keeps one workgroup per tile, =2 forces it for any count)
    const char *sk_env = getenv("SCDA_GEMM_X9_SK");
    const int sk_mode = sk_env ? atoi(sk_env) : 1;
    const size_t slot_bytes = (size_t)2 * n_cu * X::BM * X::BN * sizeof(float);
    const bool stream = !getenv("SCDA_GEMM_X9_SPLITS") && ws_bytes >= slot_bytes && (sk_mode == 2 || (sk_mode == 1 && tiles > n_cu));
    // split-K (one workgroup per (tile, split)): fill the CUs when there are fewer tiles than CUs (>= 32 slabs per split);
    // SCDA_GEMM_X9_SPLITS forces a count
    int splits = 1;
    if (const char *f = getenv("SCDA_GEMM_X9_SPLITS")) splits = atoi(f);
    else if (!stream && tiles < 200) splits = (int)std::min<long long>((256 + tiles / 2) / tiles, std::max(1, K / BK / 32));
    if (splits < 1) splits = 1;
    if (ldc != N) splits = 1;
    while (splits > 1 && (size_t)splits * M * N * sizeof(float) > ws_bytes) --splits;
    GemmGeom g{M, N, K, lda, ldb, ldc, round_k_per_split(K, splits), zero_page(), nx, ny, xcd_swizzle_enabled()};
    if (!g.zp) { set_error("scda_gemm_hip: could not allocate the zero page"); return SCDA_ELAUNCH; }
    splits = cdiv(K, g.k_per_split);
    static const bool no_mpart = getenv("SCDA_GEMM_NO_MPART") != nullptr;
    if (!no_mpart && g.swz && g.ny >= 16 && (long long)g.nx * g.ny >= 1024 && (double)M * K * sizeof(float) > 4e6) g.swz |= 2;
    Epi e{C, (float *)ws, bias, bias_on_n, act, slope, splits, accumulate, nullptr, 0.f};
    X9Stream sk{0, K / BK, 0, n_cu, 0, (float *)ws};
    if (stream) {
        sk.sk = 1;
        sk.U = tiles * sk.spt;
        sk.R = (int)((sk.U + n_cu - 1) / n_cu);
    }
    dim3 grid(stream ? (unsigned)n_cu : (unsigned)(tiles * splits));
    note_plan(X::BM, X::BN, stream ? -1 : splits, 2);      // [3] = 2: the bf16 x 9 kernel ran; [2] = -1: as a stream-K launch
    prof_begin(PK_GEMM, 2.0 * M * (double)N * K, st);
#define X9_LAUNCH(TA_, TB_)                                                                                                       \
    do {                                                                                                                          \
        if (stream) hipLaunchKernelGGL((gemm_x9_kernel<TA_, TB_, true>), grid, dim3(X::THREADS), 0, st, A, B, g, e, sk);          \
        else hipLaunchKernelGGL((gemm_x9_kernel<TA_, TB_, false>), grid, dim3(X::THREADS), 0, st, A, B, g, e, sk);                \
    } while (0)
    if (!trans_a && !trans_b) X9_LAUNCH(false, false);
    else if (!trans_a && trans_b) X9_LAUNCH(false, true);
    else if (trans_a && !trans_b) X9_LAUNCH(true, false);
    else X9_LAUNCH(true, true);
#undef X9_LAUNCH
    prof_end(st);
    int rc = launch_status("gemm_x9_kernel");
    if (rc) return rc;
    if (stream) {
        if (sk.U % sk.R == 0 && sk.R % sk.spt == 0) return SCDA_OK;      // every range is whole tiles: nothing was cut
        hipLaunchKernelGGL(gemm_x9_fixup_kernel, dim3((unsigned)(n_cu - 1), 4), dim3(256), 0, st, g, e, sk);
        return launch_status("gemm_x9_fixup_kernel");
    }
    if (splits == 1) return rc;
    return launch_dense_reduce((const float *)ws, splits, (long long)M * N, N, bias, bias_on_n, act, slope, accumulate, C, nullptr, nullptr, 0, 0, st);
}

// may this product run on the direct-to-LDS GEMM kernels (whole 16-deep slabs, 16-byte addressable rows, 32-bit lane offsets)?
static bool gemm_glds_operands_ok(const float *A, const float *B, int M, int N, int K, int lda, int ldb, int trans_a, int trans_b) {
    return (K % BK) == 0 && (lda % 4) == 0 && (ldb % 4) == 0 && ((((uintptr_t)A) | ((uintptr_t)B)) & 15) == 0 &&
           (!trans_a || (M % 4) == 0) && (!trans_b || (N % 4) == 0) && (long long)lda * 1024 + (long long)M * 4 < (1LL << 31) &&
           (long long)ldb * 1024 + (long long)N * 4 < (1LL << 31);
}

// should it take the exact-product bf16 x 9 kernel?  SCDA_GEMM_X9: 0 never, 2 whatever its size (tests), default: the FC-sized ones
static bool gemm_x9_wanted(int M, int N, int K) {
    if (getenv("SCDA_PLAN_FORCE")) return false;      // a forced (tile, split) plan names an fp32-MFMA instantiation: that one runs
    const char *x9_env = getenv("SCDA_GEMM_X9");      // (read per call: tests switch it)
    const int x9_mode = x9_env ? atoi(x9_env) : 1;
    return x9_mode == 2 || (x9_mode == 1 && M >= 256 && N >= 128 && K >= 256 && (double)M * N * K >= 4e9);
}

// A batch-1 1x1 convolution IS a dense GEMM on the NCHW tensors as they lie: Y[Cout][HW] = W[Cout][Cin] X[Cin][HW] (and its two
// gradients likewise).  The ones that are large in every dimension -- the ResNet-50 C4 detector's layer3 / RoI-head bottlenecks
// (models/mask_rcnn/resnet.py:111-148: 1024 <-> 512 <-> 2048 channels on 25088 stacked pixels) -- take the bf16 x 9 kernel; returns
// +1 (not a status code) when the call is not one of them (the caller goes on to the convolution kernels).  SCDA_CONV1X1_X9=0 turns the routing off.
static int conv1x1_as_x9(const float *A, const float *B, float *C, int M, int N, int K, int lda, int ldb, int trans_a, int trans_b,
                         const float *bias, int act, float slope, int accumulate, void *ws, size_t ws_bytes, hipStream_t st) {
    static const bool off = [] { const char *v = getenv("SCDA_CONV1X1_X9"); return v && atoi(v) == 0; }();
    if (off || !ws || (((uintptr_t)C) & 15) || !gemm_glds_operands_ok(A, B, M, N, K, lda, ldb, trans_a, trans_b) || !gemm_x9_wanted(M, N, K)) return 1;
    return gemm_x9_launch(A, B, C, M, N, K, lda, ldb, N, trans_a, trans_b, bias, 0, act, slope, accumulate, ws, ws_bytes, st);
}

// C[M][N] (ldc) = op(A) op(B) (+bias) -> act ; trans_a: A stored [K][M]; trans_b: B stored [K][N]
SCDA_API int scda_gemm_hip(const float *A, const float *B, float *C, int M, int N, int K, int lda, int ldb, int ldc,
                           int trans_a, int trans_b, const float *bias, int bias_on_n, int act, float slope,
                           int accumulate, void *ws, size_t ws_bytes, void *stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) { set_error("scda_gemm_hip: bad arguments"); return SCDA_EINVAL; }
    hipStream_t st = as_stream(stream);
    // direct-to-LDS kernel: whole 16-deep slabs, 16-byte addressable rows
    static const bool no_glds = getenv("SCDA_GEMM_NO_GLDS") != nullptr;   // A/B knob
    // (+ 32-bit lane offsets inside a descriptor: 256 rows of either operand stay below 2 GB)
    const bool glds = !no_glds && gemm_glds_operands_ok(A, B, M, N, K, lda, ldb, trans_a, trans_b);
    // exact-product bf16 x 9 form (gemm_x9_kernel): the FC-sized products; SCDA_GEMM_X9=0 keeps them on the fp32 MFMA
    if (glds && gemm_x9_wanted(M, N, K))
        return gemm_x9_launch(A, B, C, M, N, K, lda, ldb, ldc, trans_a, trans_b, bias, bias_on_n, act, slope, accumulate, ws, ws_bytes, st);
    const char *fbm = getenv("SCDA_CONV_BM");                              // 256 forces the 8-wave tile where legal
    // (not for the [K][M] x [K][N] form -- the FC weight gradient: measured 112 vs 115 TFLOP/s)
    const bool bm256_ok = glds && (M % 256) == 0 && N > 64 && !(trans_a && trans_b);
    // (FC6 dgrad: 784 128-wide tiles = 3.06 per CU, the busiest CU carries 4 -> the plan takes another tile shape)
    LaunchPlan plan = plan_launch(M, N, K, (M <= 64) ? 64 : 128, true, N > 64, false, ldc == N ? ws_bytes : 0, BK, false,
                                  bm256_ok && !(fbm && atoi(fbm) != 256));
    if (bm256_ok && fbm && atoi(fbm) == 256) { plan.bm = 256; plan.bn = 128; }
    const int BMv = plan.bm;
    const int BNv = plan.bn;
    int splits = plan.splits;
    if (splits > 1 && ldc != N) { set_error("scda_gemm_hip: split-K needs ldc == N"); return SCDA_EINVAL; }
    GemmGeom g{M, N, K, lda, ldb, ldc, round_k_per_split(K, splits), zero_page(), cdiv(N, BNv), cdiv(M, BMv),
               xcd_swizzle_enabled()};
    if (!g.zp) { set_error("scda_gemm_hip: could not allocate the zero page"); return SCDA_ELAUNCH; }
    splits = cdiv(K, g.k_per_split);
    static const bool no_mpart = getenv("SCDA_GEMM_NO_MPART") != nullptr;   // A/B knob
    // the A operand does not fit one XCD's L2 but a quarter of it does: grouped tile order (see tile_coords)
    if (!no_mpart && g.swz && g.ny >= 16 && (long long)g.nx * g.ny >= 1024 && (double)M * K * sizeof(float) > 4e6)
        g.swz |= 2;
    Epi e{C, (float *)ws, bias, bias_on_n, act, slope, splits, accumulate, nullptr, 0.f};
    dim3 grid((unsigned)g.nx * g.ny * splits);
#define GEMM_LAUNCH(BM_, BN_)                                                                            \
    do {                                                                                                 \
        if (!trans_a && !trans_b) hipLaunchKernelGGL((gemm_kernel<BM_, BN_, false, false>), grid, dim3(256), 0, st, A, B, g, e); \
        else if (!trans_a && trans_b) hipLaunchKernelGGL((gemm_kernel<BM_, BN_, false, true>), grid, dim3(256), 0, st, A, B, g, e); \
        else if (trans_a && !trans_b) hipLaunchKernelGGL((gemm_kernel<BM_, BN_, true, false>), grid, dim3(256), 0, st, A, B, g, e); \
        else hipLaunchKernelGGL((gemm_kernel<BM_, BN_, true, true>), grid, dim3(256), 0, st, A, B, g, e);  \
    } while (0)
#define GEMM_GLDS_LAUNCH(BM_, BN_)                                                                       \
    do {                                                                                                 \
        if (!trans_a && !trans_b) hipLaunchKernelGGL((gemm_glds_kernel<BM_, BN_, false, false>), grid, dim3(GemmGldsCfg<BM_, BN_>::THREADS), 0, st, A, B, g, e); \
        else if (!trans_a && trans_b) hipLaunchKernelGGL((gemm_glds_kernel<BM_, BN_, false, true>), grid, dim3(GemmGldsCfg<BM_, BN_>::THREADS), 0, st, A, B, g, e); \
        else if (trans_a && !trans_b) hipLaunchKernelGGL((gemm_glds_kernel<BM_, BN_, true, false>), grid, dim3(GemmGldsCfg<BM_, BN_>::THREADS), 0, st, A, B, g, e); \
        else hipLaunchKernelGGL((gemm_glds_kernel<BM_, BN_, true, true>), grid, dim3(GemmGldsCfg<BM_, BN_>::THREADS), 0, st, A, B, g, e);  \
    } while (0)
    note_plan(BMv, BNv, splits, glds);
    prof_begin(PK_GEMM, 2.0 * M * (double)N * K, st);
    if (glds) {
        if (BMv == 256) GEMM_GLDS_LAUNCH(256, 128);
        else if (BMv == 64 && BNv == 64) GEMM_GLDS_LAUNCH(64, 64);
        else if (BMv == 64) GEMM_GLDS_LAUNCH(64, 128);
        else if (BNv == 64) GEMM_GLDS_LAUNCH(128, 64);
        else GEMM_GLDS_LAUNCH(128, 128);
    } else if (BMv == 64 && BNv == 64) GEMM_LAUNCH(64, 64);
    else if (BMv == 64) GEMM_LAUNCH(64, 128);
    else if (BNv == 64) GEMM_LAUNCH(128, 64);
    else GEMM_LAUNCH(128, 128);
    prof_end(st);
    int rc = launch_status("gemm_kernel");
    if (rc || splits == 1) return rc;
    const long long total = (long long)M * N;
    return launch_dense_reduce((const float *)ws, splits, total, N, bias, bias_on_n, act, slope, accumulate, C, nullptr, nullptr,
                               0, 0, st);
}
