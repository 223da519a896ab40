// model.hip -- the reference's conv interpolation model, batched and fused (BASELINE config 4a).
//
// Replaces, for a batch of B grid points, what the reference does one grid point at a time:
//   CustomModel.forward            newConvolution.py:146-163  (conv1..conv4 + concatenate_bounds
//                                  :139-144 + ReLU, Flatten, linear1, tanh, linear2)
//   loss.backward()                newConvolution.py:184      (all 12 forward-path gradients)
//   nn.MSELoss / Adam step         newConvolution.py:181,185-186,224-225
//
// Algebra.  Every kernel of the conv stack has width 1, so the 6 variable columns of a sample
// never mix before linear1, and concatenate_bounds makes every conv input [start; h_c; end] for
// every channel c.  A (3,1) conv is therefore, per column n,
//     z[o,n] = b[o] + (sum_c W[o,c,0]) s[n] + (sum_c W[o,c,2]) e[n] + sum_c W[o,c,1] h[c,n]
// -- one dense (Cout x Cin) contraction plus rank-1 terms -- i.e. a chain of small GEMMs over the
// columns.  concatenate_bounds costs nothing: it is index arithmetic.
//
// Kernel.  One 512-thread workgroup owns tiles of 16 samples = 96 columns (column = w*16 + s, so
// the six columns of a sample sit 16 apart and linear1 sees one 16-column block per variable).
// Activations of every layer live in LDS; each layer is a set of 16x16 output tiles computed with
// v_mfma_f32_16x16x4_f32 (weights straight from global memory / L1, activations from LDS) and
// handed to waves round-robin.  The backward pass recomputes nothing: the forward's LDS tiles are
// still there.  Weight gradients are 74 16x16 tiles, each OWNED by one wave and accumulated in its
// registers across all the tiles the workgroup processes; they are written once per workgroup
// (float32 tile space) and a finalize kernel sums the workgroups in a fixed order in float64 and
// scatters into the PyTorch parameter layout -- deterministic, no atomics.
#include "common.h"

namespace {

constexpr int TS = 16;          // samples per tile
constexpr int NC = 6 * TS;      // columns per tile
constexpr int NP = 98;          // LDS row stride of column-indexed arrays (conflict-free transposed reads)
constexpr int SP = 18;          // LDS row stride of sample-indexed arrays
#ifndef WM_MODEL_THREADS
#define WM_MODEL_THREADS 512
#endif
constexpr int THREADS = WM_MODEL_THREADS;
constexpr int WAVES = THREADS / 64;
constexpr int NTILES = 74;      // weight-gradient tiles
constexpr int SLOTS = (NTILES + WAVES - 1) / WAVES;

static_assert(WAVES == 8, "unit and gradient-tile assignment is written for 8 waves");
constexpr int NPARAM = 20566;

// flat parameter layout = PyTorch parameter order of the forward path
constexpr int P_C1W = 0, P_C1B = 32, P_C2W = 48, P_C2B = 1584, P_C3W = 1616, P_C3B = 4688,
              P_C4W = 4720, P_C4B = 7792, P_L1W = 7824, P_L1B = 20112, P_L2W = 20176,
              P_L2B = 20560;

// LDS map (float offsets)
constexpr int L_XS = 0;                      // [2][NP]   start / end rows
constexpr int L_EE = L_XS + 2 * NP;          // [16][NP]  rows: 1, s, e, 0...
constexpr int L_H1 = L_EE + 16 * NP;         // [16][NP]
constexpr int L_H2 = L_H1 + 16 * NP;         // [32][NP]
constexpr int L_H3 = L_H2 + 32 * NP;
constexpr int L_H4 = L_H3 + 32 * NP;
constexpr int L_Z4 = L_H4 + 32 * NP;         // dL/dz of conv4 .. conv1
constexpr int L_Z3 = L_Z4 + 32 * NP;
constexpr int L_Z2 = L_Z3 + 32 * NP;
constexpr int L_Z1 = L_Z2 + 32 * NP;         // [16][NP]
constexpr int L_UU = L_Z1 + 16 * NP;         // [2][64][SP] partial linear1 sums
constexpr int L_TT = L_UU + 2 * 64 * SP;     // [64][SP]  tanh output
constexpr int L_DU = L_TT + 64 * SP;         // [64][SP]
constexpr int L_DO = L_DU + 64 * SP;         // [16][SP]  out, then dL/dout (rows >= 6 zero)
constexpr int L_E2 = L_DO + 16 * SP;         // [16][SP]  row 0 = 1, rest 0
constexpr int L_AUX = L_E2 + 16 * SP;        // [3 layers][2][32]  sum_c W[o,c,0], sum_c W[o,c,2]
constexpr int L_RED = L_AUX + 192;           // [WAVES] loss partials
// packed MFMA A operands (64 floats per operand, indexed by lane), filled once per workgroup
constexpr int W_C1F = L_RED + 16;              // conv1: 1 operand
constexpr int W_C2F = W_C1F + 64;              // conv2 fwd: 2 o-tiles x (1 + 4) steps
constexpr int W_C3F = W_C2F + 2 * 5 * 64;      // conv3 fwd: 2 x (1 + 8)
constexpr int W_C4F = W_C3F + 2 * 9 * 64;
constexpr int W_C4B = W_C4F + 2 * 9 * 64;      // conv4 bwd data: 2 c-tiles x 8 steps
constexpr int W_C3B = W_C4B + 2 * 8 * 64;
constexpr int W_C2B = W_C3B + 2 * 8 * 64;      // conv2 bwd data: 1 c-tile x 8
constexpr int W_L2F = W_C2B + 8 * 64;          // linear2 fwd: 16 steps
constexpr int W_L2B = W_L2F + 16 * 64;         // dT: 4 j-tiles x 2 steps
constexpr int L_B1 = W_L2B + 8 * 64;           // linear1 bias [64]
constexpr int L_B2 = L_B1 + 64;                // linear2 bias [16] (rows >= 6 zero)
constexpr int L_TOTAL = L_B2 + 16;
constexpr size_t LDS_BYTES = (size_t)L_TOTAL * sizeof(float);

__device__ __forceinline__ f32x4 mma(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// tanh(u) = 1 - 2 / (exp(2u) + 1) on the hardware exp2 / rcp: absolute error < 4e-7 (the libm
// tanhf took 1 k of the 28.7 k cycles of a tile); saturates correctly at +-1
__device__ __forceinline__ float fast_tanh(float u)
{
    return 1.0f - __fdividef(2.0f, __expf(2.0f * u) + 1.0f);
}

// Barrier between two phases that communicate through LDS only (no vmcnt(0) drain: loads whose
// values are consumed later stay in flight across it).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// write a 16x16 accumulator tile (row = 4g + r, col = i) into an LDS matrix
__device__ __forceinline__ void put_tile(float *dst, int stride, int row0, int col0, f32x4 acc,
                                         int i, int g)
{
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(row0 + 4 * g + r) * stride + col0 + i] = acc[r];
}

// forward conv layer: Hout[o][n] = relu(b[o] + as[o] s[n] + ae[o] e[n] + sum_c Wm[o][c] Hin[c][n])
// K = Cin (0 for conv1), O = Cout; weight element (o,c) sits at wbase[(o*K + c)*3 + 1].
// Packed operands (LDS, wp): per o-tile 1 + K/4 operands; operand 0 = [b, as, ae, 0] by lane group.
// The conv stack is column-local (every kernel has width 1): column tile n0 of a layer's output
// needs only column tile n0 of its input.  Wave w < 6 therefore owns column tile w through conv1
// .. conv4 (and back through conv4 .. conv2 in the backward pass) with no workgroup barrier in
// between -- its own LDS writes are read back by its own lanes, in program order -- and runs the
// two 16-row output tiles of a layer as two interleaved MFMA chains (each unit is one dependent
// chain on a single accumulator and the phases are latency-bound: SQ counters showed the MFMA
// pipe 41 % busy with the waves waiting 56 % of the time).
// The MFMA's contraction index is the lane group g, and nothing obliges step t to take input
// channels in order: step (T, r) takes channel 16 T + 4 g + r -- which is exactly the element
// acc_T[r] this lane holds of the layer below (an accumulator tile keeps row 4 g + r, column i in
// lane (i, g)).  So a layer's B operands ARE the previous layer's accumulators, in registers: no
// LDS write -> read round trip between the layers of a wave's chain (the tiles still go to LDS for
// the backward pass, but nothing waits for them).  The packed weights follow the same order.
template <int K>   // K = Cin; Cout = 32
__device__ __forceinline__ void conv_fwd_col(float *lds, int wp, const f32x4 (&hin)[K / 16],
                                             f32x4 (&hout)[2], int hout_off, int c0, int lane, int i,
                                             int g)
{
    constexpr int STEPS = K / 4 + 1;
    const float *a0 = lds + wp + lane, *a1 = a0 + STEPS * 64;
    // rank-1 terms and bias as one MFMA step: A = [b, as, ae, 0], B = [1, s, e, 0]
    const float ee = lds[L_EE + g * NP + c0 + i];         // EE rows 0..3 = 1, s, e, 0
    f32x4 acc0 = mma(a0[0], ee, (f32x4){0.f, 0.f, 0.f, 0.f});
    f32x4 acc1 = mma(a1[0], ee, (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
    for (int t = 0; t < K / 4; ++t) {
        const float b = hin[t >> 2][t & 3];
        acc0 = mma(a0[(t + 1) * 64], b, acc0);
        acc1 = mma(a1[(t + 1) * 64], b, acc1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc0[r] = fmaxf(acc0[r], 0.0f); acc1[r] = fmaxf(acc1[r], 0.0f); }
    put_tile(lds + hout_off, NP, 0, c0, acc0, i, g);
    put_tile(lds + hout_off, NP, 16, c0, acc1, i, g);
    hout[0] = acc0;
    hout[1] = acc1;
}

// backward-data of a conv layer for one column tile:
// dZin[c][n] = relu'(Hin[c][n]) * sum_o Wm[o][c] dZout[o][n];  K = Cin (16 or 32 rows), Cout = 32
// The contraction over the 32 output channels takes them in the same order (step (T, r) <-> channel
// 16 T + 4 g + r), so dZout comes from the registers of the layer above as well.
template <int K>
__device__ __forceinline__ void conv_bwd_col(float *lds, int wp, const f32x4 (&zout)[2], int hin,
                                             int zin_off, f32x4 (&zin)[2], int n0, int lane, int i,
                                             int g)
{
    const float *a0 = lds + wp + lane, *a1 = a0 + 8 * 64;
    // relu' mask of the layer below: read ahead of the chain
    float hm0[4], hm1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        hm0[r] = lds[hin + (4 * g + r) * NP + n0 + i];
        hm1[r] = K == 32 ? lds[hin + (16 + 4 * g + r) * NP + n0 + i] : 0.0f;
    }
    f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const float b = zout[t >> 2][t & 3];
        acc0 = mma(a0[t * 64], b, acc0);
        if (K == 32) acc1 = mma(a1[t * 64], b, acc1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        acc0[r] = hm0[r] > 0.0f ? acc0[r] : 0.0f;
        if (K == 32) acc1[r] = hm1[r] > 0.0f ? acc1[r] : 0.0f;
    }
    put_tile(lds + zin_off, NP, 0, n0, acc0, i, g);
    if (K == 32) put_tile(lds + zin_off, NP, 16, n0, acc1, i, g);
    zin[0] = acc0;
    zin[1] = acc1;
}

// fill the packed operands of a forward conv layer (see conv_fwd_col)
template <int K>
__device__ __forceinline__ void pack_conv_fwd(float *lds, int wp, const float *__restrict__ wmid,
                                              const float *__restrict__ bias, int aux, int tid)
{
    constexpr int STEPS = K / 4 + 1;
    for (int e = tid; e < 2 * STEPS * 64; e += THREADS) {
        const int lane = e & 63, q = e >> 6, step = q % STEPS, ot = q / STEPS;
        const int i = lane & 15, g = lane >> 4, o = ot * 16 + i;
        float v;
        if (step == 0)
            v = g == 0 ? bias[o] : (g == 1 ? lds[aux + o] : (g == 2 ? lds[aux + 32 + o] : 0.0f));
        else
            v = wmid[(o * K + ((step - 1) >> 2) * 16 + 4 * g + ((step - 1) & 3)) * 3 + 1];
        lds[wp + e] = v;
    }
}

// packed operands of a backward-data conv layer: [c-tile][step t] -> W[o = 16 (t/4) + 4 g + t%4][c]
template <int K, int O>
__device__ __forceinline__ void pack_conv_bwd(float *lds, int wp, const float *__restrict__ wmid,
                                              int tid)
{
    for (int e = tid; e < (K / 16) * (O / 4) * 64; e += THREADS) {
        const int lane = e & 63, q = e >> 6, t = q % (O / 4), ct = q / (O / 4);
        const int i = lane & 15, g = lane >> 4;
        lds[wp + e] = wmid[(((t >> 2) * 16 + 4 * g + (t & 3)) * K + ct * 16 + i) * 3 + 1];
    }
}

// weight-gradient tile: C[a][b] += sum_k A[arow0 + a][ka0 + k] * B[brow0 + b][kb0 + k]
struct GradTile {
    int a_off, a_stride, b_off, b_stride, ka0, kb0, steps;
};

__device__ __forceinline__ GradTile grad_tile(int id)
{
    GradTile t;
    t.a_stride = NP; t.b_stride = NP; t.ka0 = 0; t.kb0 = 0; t.steps = NC / 4;
    if (id < 17) {
        // conv layers: per layer [main tiles ..., extras ...]
        int zoff, hoff, lid = id;
        if (id < 6)       { zoff = L_Z4; hoff = L_H3; }
        else if (id < 12) { zoff = L_Z3; hoff = L_H2; lid = id - 6; }
        else if (id < 16) { zoff = L_Z2; hoff = L_H1; lid = id - 12; }
        else              { zoff = L_Z1; hoff = L_EE; lid = 0; }
        if (id < 12) {           // conv4 / conv3: 4 main (ot*2+ct), 2 extras (ot)
            if (lid < 4) { t.a_off = zoff + (lid >> 1) * 16 * NP; t.b_off = hoff + (lid & 1) * 16 * NP; }
            else         { t.a_off = zoff + (lid - 4) * 16 * NP; t.b_off = L_EE; }
        } else if (id < 16) {    // conv2: 2 main (ot), 2 extras (ot)
            if (lid < 2) { t.a_off = zoff + lid * 16 * NP; t.b_off = hoff; }
            else         { t.a_off = zoff + (lid - 2) * 16 * NP; t.b_off = L_EE; }
        } else {                 // conv1: extras only
            t.a_off = zoff; t.b_off = L_EE;
        }
    } else if (id < 65) {        // linear1 main: 17 + (jt*6 + w)*2 + ct
        const int q = id - 17, ct = q & 1, jw = q >> 1, w = jw % 6, jt = jw / 6;
        t.a_off = L_DU + jt * 16 * SP; t.a_stride = SP; t.ka0 = 0;
        t.b_off = L_H4 + ct * 16 * NP; t.b_stride = NP; t.kb0 = w * 16;
        t.steps = TS / 4;
    } else if (id < 69) {        // linear1 bias (jt)
        t.a_off = L_DU + (id - 65) * 16 * SP; t.a_stride = SP;
        t.b_off = L_E2; t.b_stride = SP; t.steps = TS / 4;
    } else if (id < 73) {        // linear2 main (jt): C[m][j]
        t.a_off = L_DO; t.a_stride = SP;
        t.b_off = L_TT + (id - 69) * 16 * SP; t.b_stride = SP; t.steps = TS / 4;
    } else {                     // linear2 bias
        t.a_off = L_DO; t.a_stride = SP; t.b_off = L_E2; t.b_stride = SP; t.steps = TS / 4;
    }
    return t;
}

// Slot s of wave w -> weight-gradient tile id (or -1).  Tiles 0..16 run 24 MFMA steps per sample
// tile, tiles 17..73 four: wave 0 takes three long tiles and two short ones, waves 1..7 two long
// and up to eight short ones -- 76..80 steps each instead of 76..100 with a round-robin deal.
__device__ __forceinline__ int grad_tile_of(int s, int w)
{
    if (s < 2) return s * 8 + w;
    if (w == 0) return s == 2 ? 16 : (s < 5 ? 21 + s : -1);          // 16, 24, 25
    if (s == 2) return 16 + w;                                        // 17..23
    const int id = 26 + (s - 3) * 7 + (w - 1);
    return id < NTILES ? id : -1;
}

#ifdef WM_MODEL_STAMPS
// debug build only (tools/probe/model_stamps.py): cycle stamps of wave 0 of workgroup 0 at the
// phase boundaries of its 10th tile
__device__ unsigned long long wm_stamps[32];
#define STAMP(k)                                                                     \
    do {                                                                             \
        if (blockIdx.x == 0 && tid == 0 && tile == (long long)blockIdx.x + 10ll * gridDim.x) \
            wm_stamps[k] = clock64();                                                \
    } while (0)
extern "C" int wm_debug_read_stamps(unsigned long long *dst)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(wm_stamps), sizeof(unsigned long long) * 32);
}
#else
#define STAMP(k)
#endif

// Where a GATHER launch takes its samples from: the resident (T,Z,Y,X,6) volume and the index
// decode of CustomDataset.__getitem__ (newConvolution.py:93-106), i.e. what wm_gather_triplets_f32
// would have written to x and y first (144 B per sample each way).
struct ModelGather {
    const float *vol;
    const long long *idx;
    long long idx0;
    unsigned long long plane;                 // elements of one time slice = Z*Y*X*6
    unsigned tot, Sx, Sy, Sz, St, Y, X, time_offset, d_mid, d_end;
};

// element offset in vol of variable 0 of sample `smp` at its start time
__device__ __forceinline__ size_t gather_base(const ModelGather &gs, long long smp)
{
    unsigned id;
    if (gs.idx) {
        long long raw = gs.idx[smp] % (long long)gs.tot;
        if (raw < 0) raw += gs.tot;
        id = (unsigned)raw;
    } else {
        id = (unsigned)((unsigned long long)(gs.idx0 + smp) % gs.tot);
    }
    const unsigned q1 = id / gs.Sx, column = id - q1 * gs.Sx;
    const unsigned q2 = q1 / gs.Sy, row = q1 - q2 * gs.Sy;
    const unsigned q3 = q2 / gs.Sz, height = q2 - q3 * gs.Sz;
    const unsigned time = q3 % gs.St + gs.time_offset;
    return (size_t)time * gs.plane + ((size_t)(height * gs.Y + row) * gs.X + column) * 6;
}

// MODE 0: forward only.  MODE 1: forward + backward with dL/dout = 2 (out - y) * dscale (fused
// MSE).  MODE 2: forward + backward with dL/dout given.  GATHER (MODE 1): x and y are read from
// the volume by sample index instead of from (B,2,6) / (B,6) arrays.
template <int MODE, bool GATHER = false>
__global__ __launch_bounds__(THREADS) void model_kernel(
    const float *__restrict__ prm, const float *__restrict__ x, const float *__restrict__ y,
    const float *__restrict__ dout, float *__restrict__ out, float *__restrict__ gtiles,
    float *__restrict__ loss_part, long long B, float dscale, const float *__restrict__ l1pack,
    ModelGather gs)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;

    // per-workgroup constants: rank-1 coefficients of conv2..4, the constant rows of EE / E2
    for (int e = tid; e < 192; e += THREADS) {
        const int layer = e / 64, which = (e >> 5) & 1, o = e & 31;
        const int K = layer == 0 ? 16 : 32;
        const float *w = prm + (layer == 0 ? P_C2W : (layer == 1 ? P_C3W : P_C4W));
        float s = 0.0f;
        for (int c = 0; c < K; ++c) s += w[(o * K + c) * 3 + (which ? 2 : 0)];
        lds[L_AUX + e] = s;
    }
    for (int e = tid; e < 16 * NP; e += THREADS) lds[L_EE + e] = (e < NP) ? 1.0f : 0.0f;
    for (int e = tid; e < 16 * SP; e += THREADS) {
        lds[L_E2 + e] = (e < SP) ? 1.0f : 0.0f;
        lds[L_DO + e] = 0.0f;
    }
    f32x4 gacc[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) gacc[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float loss_acc = 0.0f;
    __syncthreads();
    // packed MFMA weight operands (the unit -> wave assignment is static, the weights are
    // constant for the launch): read from global once per workgroup, from LDS per tile
    if (tid < 64)
        lds[W_C1F + tid] = g == 0 ? prm[P_C1B + i]
                                  : (g == 1 ? prm[P_C1W + i * 2]
                                            : (g == 2 ? prm[P_C1W + i * 2 + 1] : 0.0f));
    pack_conv_fwd<16>(lds, W_C2F, prm + P_C2W, prm + P_C2B, L_AUX, tid);
    pack_conv_fwd<32>(lds, W_C3F, prm + P_C3W, prm + P_C3B, L_AUX + 64, tid);
    pack_conv_fwd<32>(lds, W_C4F, prm + P_C4W, prm + P_C4B, L_AUX + 128, tid);
    if (tid < 64) lds[L_B1 + tid] = prm[P_L1B + tid];
    if (tid < 16) lds[L_B2 + tid] = tid < 6 ? prm[P_L2B + tid] : 0.0f;
    for (int e = tid; e < 16 * 64; e += THREADS) {       // linear2 fwd: step t -> L2[i][t + 16 g]
        const int l = e & 63, t = e >> 6, li = l & 15, lg = l >> 4;
        lds[W_L2F + e] = li < 6 ? prm[P_L2W + li * 64 + t + 16 * lg] : 0.0f;
    }
    if constexpr (MODE != 0) {
        pack_conv_bwd<32, 32>(lds, W_C4B, prm + P_C4W, tid);
        pack_conv_bwd<32, 32>(lds, W_C3B, prm + P_C3W, tid);
        pack_conv_bwd<16, 32>(lds, W_C2B, prm + P_C2W, tid);
        for (int e = tid; e < 8 * 64; e += THREADS) {    // dT: [j-tile][t] -> L2[m = t + 2 g][j]
            const int l = e & 63, q = e >> 6, t = q & 1, jt = q >> 1, li = l & 15, lg = l >> 4;
            const int m = t + 2 * lg;
            lds[W_L2B + e] = m < 6 ? prm[P_L2W + m * 64 + jt * 16 + li] : 0.0f;
        }
    }
    __syncthreads();

    // linear1 operands (48 values per lane: the forward unit of this wave (j-tile, w-half) and its
    // three backward half-units (c-tile, w, j-half)).  Forward-only launches keep the forward set
    // in registers.  Training launches do not: with the 40 gradient accumulators the 48 values
    // did not fit the 256 registers of a wave (23 spilled, and every reload at the head of a
    // phase exposed a scratch round trip: 1.5 k of the 28 k cycles of a tile in the first phase
    // alone).  They are re-read per tile from a lane-ordered copy in the workspace (l1pack, 96
    // KB, L2-resident, six 16-byte loads per set) one phase before they are needed.
    float l1f[24];
    if constexpr (MODE == 0) {
        const int jt = wave >> 1, half = wave & 1, j = jt * 16 + i;
#pragma unroll
        for (int ww = 0; ww < 3; ++ww)
#pragma unroll
            for (int t = 0; t < 8; ++t)
                l1f[ww * 8 + t] = prm[P_L1W + j * 192 + (t + 8 * g) * 6 + half * 3 + ww];
    }
    f32x4 lf[6], lb[8];
    (void)lf; (void)lb;

    const long long ntiles = (B + TS - 1) / TS;
    // x of the NEXT tile is fetched while the current one is computed (one float per thread)
    const int xr = tid / NC, xcol = tid % NC, xoff = xr * 6 + (xcol >> 4), xs = xcol & 15;
    // every load below is unconditional on a clamped index and its validity is applied where the
    // value is used: a data-dependent branch around a load makes hipcc drain vmcnt(0) at the join
    // (1.8 k cycles of HBM latency per tile, measured with cycle stamps)
    float xnext = 0.0f;
    bool xvalid = false;
    // GATHER: row xr of x is the slice d_end * xr after the sample's start time, variable xcol >> 4
    const size_t xgoff = GATHER ? (size_t)xr * gs.d_end * gs.plane + (size_t)(xcol >> 4) : 0;
    if (tid < 2 * NC && (long long)blockIdx.x < ntiles) {
        const long long smp = (long long)blockIdx.x * TS + xs;
        xvalid = smp < B;
        const long long sc = smp < B ? smp : B - 1;
        if constexpr (GATHER) xnext = gs.vol[gather_base(gs, sc) + xgoff];
        else xnext = x[sc * 12 + xoff];
    }
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long s0 = tile * TS;
        STAMP(0);
        // ---- x: XS[r][w*16+s] = x[s][r][w]; EE rows 1,2 mirror it
        if (tid < 2 * NC) {
            const float xv = xvalid ? xnext : 0.0f;
            lds[L_XS + xr * NP + xcol] = xv;
            lds[L_EE + (1 + xr) * NP + xcol] = xv;
            const long long smp = (tile + gridDim.x) * TS + xs;
            xvalid = tile + gridDim.x < ntiles && smp < B;
            const long long sc = smp < B ? smp : B - 1;
            if constexpr (GATHER) xnext = gs.vol[gather_base(gs, sc) + xgoff];
            else xnext = x[sc * 12 + xoff];
        }
        if constexpr (MODE != 0) {
            // opaque per-tile copy of the lane offset: otherwise the loads are loop-invariant and
            // hipcc hoists them back into 24 registers for the whole kernel (the base stays a
            // kernel argument so that these are global, not flat, loads)
            unsigned lo = (unsigned)((wave * 14 * 64 + lane) * 16);
            asm volatile("" : "+v"(lo));
#pragma unroll
            for (int q = 0; q < 6; ++q) lf[q] = *(const f32x4 *)((const char *)l1pack + lo + q * 1024);
        }
        // targets (or upstream gradient) of this tile, consumed by wave 0 after linear2 (only for
        // m < 6 and samples < B; the rest are reads of clamped entries that are never used)
        float tgt[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (MODE != 0) {
            if (wave == 0) {
                const long long smp = s0 + i < B ? s0 + i : B - 1;
                size_t ybase = 0;
                if constexpr (GATHER) ybase = gather_base(gs, smp) + (size_t)gs.d_mid * gs.plane;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 4 * g + r < 6 ? 4 * g + r : 5;
                    if constexpr (GATHER) tgt[r] = gs.vol[ybase + m];
                    else tgt[r] = MODE == 1 ? y[smp * 6 + m] : dout[smp * 6 + m];
                }
            }
        }
        lds_barrier();        // the x of the next tile and this tile's targets stay in flight
        STAMP(1);
        // ---- forward
        // conv1 weight (16,1,2,1): as = W[o][0][0], ae = W[o][0][1]  (strided view)
        if (wave < 6) {
            const int c0 = wave * 16;
            const float a0 = lds[W_C1F + lane];
            const float b0 = lds[L_EE + g * NP + c0 + i];
            f32x4 acc = mma(a0, b0, (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = fmaxf(acc[r], 0.0f);
            put_tile(lds + L_H1, NP, 0, c0, acc, i, g);
            f32x4 h1[1] = {acc}, h2[2], h3[2], h4[2];
            conv_fwd_col<16>(lds, W_C2F, h1, h2, L_H2, c0, lane, i, g);
            conv_fwd_col<32>(lds, W_C3F, h2, h3, L_H3, c0, lane, i, g);
            conv_fwd_col<32>(lds, W_C4F, h3, h4, L_H4, c0, lane, i, g);
        }
        lds_barrier();
        STAMP(2);
        // linear1: u[j][s] = sum_w sum_c L1[j][c*6+w] H4[c][w*16+s]; 8 units = 4 j-tiles x 2 w-halves
        {
            const int jt = wave >> 1, half = wave & 1;
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ww = 0; ww < 3; ++ww) {
                const int w = half * 3 + ww;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int c = t + 8 * g, e = ww * 8 + t;
                    const float a = MODE == 0 ? l1f[e] : lf[e >> 2][e & 3];
                    acc = mma(a, lds[L_H4 + c * NP + w * 16 + i], acc);
                }
            }
            put_tile(lds + L_UU + half * 64 * SP, SP, jt * 16, 0, acc, i, g);
        }
        if constexpr (MODE != 0) {
            // the backward set, needed four phases from here
            unsigned lo = (unsigned)(((wave * 14 + 6) * 64 + lane) * 16);
            asm volatile("" : "+v"(lo));
#pragma unroll
            for (int q = 0; q < 8; ++q) lb[q] = *(const f32x4 *)((const char *)l1pack + lo + q * 1024);
        }
        lds_barrier();
        STAMP(3);
        for (int e = tid; e < 64 * TS; e += THREADS) {
            const int j = e >> 4, s = e & 15;
            const float u = lds[L_UU + j * SP + s] + lds[L_UU + 64 * SP + j * SP + s] + lds[L_B1 + j];
            lds[L_TT + j * SP + s] = fast_tanh(u);
        }
        lds_barrier();
        STAMP(4);
        // linear2: out[m][s] = bl2[m] + sum_j L2[m][j] T[j][s]   (one unit; rows m >= 6 are zero)
        if (wave == 0) {
            // four independent chains of four steps (one chain of 16 dependent MFMAs kept the
            // other seven waves waiting 2.8 k cycles per tile), summed pairwise
            f32x4 q0 = (f32x4){0.f, 0.f, 0.f, 0.f}, q1 = q0, q2 = q0, q3 = q0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                q0 = mma(lds[W_L2F + t * 64 + lane], lds[L_TT + (t + 16 * g) * SP + i], q0);
                q1 = mma(lds[W_L2F + (t + 4) * 64 + lane], lds[L_TT + (t + 4 + 16 * g) * SP + i], q1);
                q2 = mma(lds[W_L2F + (t + 8) * 64 + lane], lds[L_TT + (t + 8 + 16 * g) * SP + i], q2);
                q3 = mma(lds[W_L2F + (t + 12) * 64 + lane], lds[L_TT + (t + 12 + 16 * g) * SP + i], q3);
            }
            const f32x4 acc = (q0 + q1) + (q2 + q3);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 4 * g + r;
                const long long smp = s0 + i;
                float o = 0.0f;
                if (m < 6) {
                    o = acc[r] + lds[L_B2 + m];
                    if (smp < B) out[smp * 6 + m] = o;
                }
                if constexpr (MODE != 0) {
                    float d = 0.0f;
                    if (m < 6 && smp < B) {
                        if constexpr (MODE == 1) {
                            const float diff = o - tgt[r];
                            loss_acc += diff * diff;
                            d = 2.0f * diff * dscale;
                        } else {
                            d = tgt[r];
                        }
                    }
                    lds[L_DO + m * SP + i] = d;
                }
            }
        }
        if constexpr (MODE == 0) {
            lds_barrier();
        STAMP(5);
            continue;
        }
        lds_barrier();
        STAMP(6);
        // ---- backward (data path)
        // dT[j][s] = sum_m L2[m][j] dO[m][s];  dU = dT (1 - T^2): 4 units (j-tiles)
        if (wave < 4) {
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 2; ++t) {           // m = 0..7 (rows >= 6 of dO are zero)
                const int m = t + 2 * g;
                acc = mma(lds[W_L2B + (wave * 2 + t) * 64 + lane], lds[L_DO + m * SP + i], acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float tv = lds[L_TT + (wave * 16 + 4 * g + r) * SP + i];
                acc[r] *= (1.0f - tv * tv);
            }
            put_tile(lds + L_DU, SP, wave * 16, 0, acc, i, g);
        }
        lds_barrier();
        STAMP(7);
        // dH4[c][w*16+s] = relu'(H4) * sum_j L1[j][c*6+w] dU[j][s]: like the conv layers this is
        // local to column tile w, so wave w < 6 runs its two c-tiles as two interleaved chains
        // (they share the B operand) and goes straight on into the conv backward stack -- no
        // half-unit sums, no mask pass, no barrier (those two phases took 3.7 k cycles of 26 k)
        if (wave < 6) {
            const int n0 = wave * 16;
            f32x4 z4[2], z3[2], z2[2], z1[2];
            {
                const float *du = lds + L_DU + 16 * g * SP + i;
                f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const float b = du[t * SP];
                    acc0 = mma(lb[t >> 2][t & 3], b, acc0);
                    acc1 = mma(lb[4 + (t >> 2)][t & 3], b, acc1);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc0[r] = lds[L_H4 + (4 * g + r) * NP + n0 + i] > 0.0f ? acc0[r] : 0.0f;
                    acc1[r] = lds[L_H4 + (16 + 4 * g + r) * NP + n0 + i] > 0.0f ? acc1[r] : 0.0f;
                }
                put_tile(lds + L_Z4, NP, 0, n0, acc0, i, g);
                put_tile(lds + L_Z4, NP, 16, n0, acc1, i, g);
                z4[0] = acc0;
                z4[1] = acc1;
            }
            conv_bwd_col<32>(lds, W_C4B, z4, L_H3, L_Z3, z3, n0, lane, i, g);
            conv_bwd_col<32>(lds, W_C3B, z3, L_H2, L_Z2, z2, n0, lane, i, g);
            conv_bwd_col<16>(lds, W_C2B, z2, L_H1, L_Z1, z1, n0, lane, i, g);
        }
        lds_barrier();
        STAMP(10);
        // ---- weight gradients: every tile owned by one wave, accumulated in registers
        // slots in pairs: two tiles with the same step count run as two interleaved chains
#pragma unroll
        for (int s = 0; s < SLOTS; s += 2) {
            const int id0 = grad_tile_of(s, wave);
            const int id1 = s + 1 < SLOTS ? grad_tile_of(s + 1, wave) : -1;
            if (id0 >= 0 && id1 >= 0) {
                const GradTile t0 = grad_tile(id0), t1 = grad_tile(id1);
                const float *ap0 = lds + t0.a_off + i * t0.a_stride + t0.ka0 + g;
                const float *bp0 = lds + t0.b_off + i * t0.b_stride + t0.kb0 + g;
                const float *ap1 = lds + t1.a_off + i * t1.a_stride + t1.ka0 + g;
                const float *bp1 = lds + t1.b_off + i * t1.b_stride + t1.kb0 + g;
                f32x4 acc0 = gacc[s], acc1 = gacc[s + 1 < SLOTS ? s + 1 : s];
                if (t0.steps == t1.steps) {
#pragma unroll 4
                    for (int k = 0; k < t0.steps; ++k) {
                        acc0 = mma(ap0[4 * k], bp0[4 * k], acc0);
                        acc1 = mma(ap1[4 * k], bp1[4 * k], acc1);
                    }
                } else {
#pragma unroll 4
                    for (int k = 0; k < t0.steps; ++k) acc0 = mma(ap0[4 * k], bp0[4 * k], acc0);
#pragma unroll 4
                    for (int k = 0; k < t1.steps; ++k) acc1 = mma(ap1[4 * k], bp1[4 * k], acc1);
                }
                gacc[s] = acc0;
                if (s + 1 < SLOTS) gacc[s + 1] = acc1;
            } else if (id0 >= 0) {
                const GradTile t = grad_tile(id0);
                const float *ap = lds + t.a_off + i * t.a_stride + t.ka0 + g;
                const float *bp = lds + t.b_off + i * t.b_stride + t.kb0 + g;
                f32x4 acc = gacc[s];
#pragma unroll 4
                for (int k = 0; k < t.steps; ++k) acc = mma(ap[4 * k], bp[4 * k], acc);
                gacc[s] = acc;
            }
        }
        lds_barrier();
        STAMP(11);
    }

    if constexpr (MODE != 0) {
        // tile space out: [block][tile id][r*64 + lane]
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int id = grad_tile_of(s, wave);
            if (id >= 0) {
                float *dst = gtiles + ((size_t)blockIdx.x * NTILES + id) * 256;
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[r * 64 + lane] = gacc[s][r];
            }
        }
        if constexpr (MODE == 1) {
            const float ls = wave_sum(loss_acc);     // only wave 0 holds non-zero sums
            if (tid == 0) loss_part[blockIdx.x] = ls;
        }
    }
}

// tile element (row R, col C) -> index inside a tile's 256 floats
__device__ __forceinline__ int tile_elem(int R, int C) { return (R & 3) * 64 + (R >> 2) * 16 + C; }

// parameter index -> (tile id, element)
__device__ __forceinline__ void param_src(int p, int &id, int &el)
{
    if (p < P_C1B) {                       // conv1.weight (16,1,2,1): o*2 + kh -> extras col 1 + kh
        id = 16; el = tile_elem(p >> 1, 1 + (p & 1));
    } else if (p < P_C2W) {                // conv1.bias
        id = 16; el = tile_elem(p - P_C1B, 0);
    } else if (p < P_C2B) {                // conv2.weight (32,16,3,1)
        const int q = p - P_C2W, kh = q % 3, c = (q / 3) % 16, o = q / 48;
        if (kh == 1) { id = 12 + (o >> 4); el = tile_elem(o & 15, c); }
        else         { id = 14 + (o >> 4); el = tile_elem(o & 15, kh == 0 ? 1 : 2); }
    } else if (p < P_C3W) {                // conv2.bias
        const int o = p - P_C2B; id = 14 + (o >> 4); el = tile_elem(o & 15, 0);
    } else if (p < P_C3B || (p >= P_C4W && p < P_C4B)) {   // conv3 / conv4 weight (32,32,3,1)
        const bool l4 = p >= P_C4W;
        const int q = p - (l4 ? P_C4W : P_C3W), kh = q % 3, c = (q / 3) % 32, o = q / 96;
        const int base = l4 ? 0 : 6;
        if (kh == 1) { id = base + (o >> 4) * 2 + (c >> 4); el = tile_elem(o & 15, c & 15); }
        else         { id = base + 4 + (o >> 4); el = tile_elem(o & 15, kh == 0 ? 1 : 2); }
    } else if (p < P_C4W) {                // conv3.bias
        const int o = p - P_C3B; id = 10 + (o >> 4); el = tile_elem(o & 15, 0);
    } else if (p < P_L1W) {                // conv4.bias
        const int o = p - P_C4B; id = 4 + (o >> 4); el = tile_elem(o & 15, 0);
    } else if (p < P_L1B) {                // linear1.weight (64,192): j*192 + c*6 + w
        const int q = p - P_L1W, j = q / 192, cw = q % 192, c = cw / 6, w = cw % 6;
        id = 17 + ((j >> 4) * 6 + w) * 2 + (c >> 4); el = tile_elem(j & 15, c & 15);
    } else if (p < P_L2W) {                // linear1.bias
        const int j = p - P_L1B; id = 65 + (j >> 4); el = tile_elem(j & 15, 0);
    } else if (p < P_L2B) {                // linear2.weight (6,64): m*64 + j
        const int q = p - P_L2W, m = q >> 6, j = q & 63;
        id = 69 + (j >> 4); el = tile_elem(m, j & 15);
    } else {                               // linear2.bias
        id = 73; el = tile_elem(p - P_L2B, 0);
    }
}

constexpr int GSPLIT = 4;      // workgroup ranges summed separately, then combined in order

// stage 1: tile-space sums over the workgroups, coalesced (thread = one tile element).
// tsum: [GSPLIT][NTILES][256] float64.
__global__ __launch_bounds__(256) void model_tile_reduce_kernel(const float *__restrict__ gtiles,
                                                                int nblocks,
                                                                double *__restrict__ tsum)
{
    const int id = blockIdx.x, part = blockIdx.y, el = threadIdx.x;
    const int b0 = (int)((long long)nblocks * part / GSPLIT);
    const int b1 = (int)((long long)nblocks * (part + 1) / GSPLIT);
    double s = 0.0;
#pragma unroll 8
    for (int b = b0; b < b1; ++b) s += (double)gtiles[((size_t)b * NTILES + id) * 256 + el];
    tsum[((size_t)part * NTILES + id) * 256 + el] = s;
}

// stage 2: tile space -> PyTorch parameter layout, partial sums combined in a fixed order
__global__ __launch_bounds__(256) void model_grad_finalize_kernel(const double *__restrict__ tsum,
                                                                  const float *__restrict__ loss_part,
                                                                  int nblocks, float *__restrict__ grads,
                                                                  float *__restrict__ loss, float loss_scale)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < NPARAM) {
        int id, el;
        param_src(p, id, el);
        double s = 0.0;
#pragma unroll
        for (int part = 0; part < GSPLIT; ++part) s += tsum[((size_t)part * NTILES + id) * 256 + el];
        grads[p] = (float)s;
    }
    if (p == 0 && loss && loss_part) {
        double s = 0.0;
        for (int b = 0; b < nblocks; ++b) s += (double)loss_part[b];
        *loss = (float)(s * (double)loss_scale);
    }
}

// torch.optim.Adam defaults semantics (newConvolution.py:225): no weight decay, no amsgrad.  The
// operation order is the single-tensor path of torch.optim.Adam: lerp for the first moment,
// mul + addcmul for the second, denom = sqrt(v) / sqrt(bc2) + eps, p += (-lr / bc1) * (m / denom);
// every scalar is computed in double on the host and rounded to float once, as torch does.
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ gr,
                                                   float *__restrict__ m, float *__restrict__ v,
                                                   long long count, AdamScalars a)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= count) return;
    float pv = p[idx], mv = m[idx], vv = v[idx];
    adam_update(pv, mv, vv, gr[idx], a);
    m[idx] = mv;
    v[idx] = vv;
    p[idx] = pv;
}

int model_grid(long long B)
{
    const long long ntiles = (B + TS - 1) / TS;
    long long g = wm_num_cus();
    if (g > ntiles) g = ntiles;
    if (g < 1) g = 1;
    return (int)g;
}

// lane-ordered copy of linear1.weight for the training kernel: [wave][14][lane][4] floats, slots
// 0..23 = the forward operands of the wave's (j-tile, w-half) unit, 24..55 = the backward operands
// of column tile w = wave (two c-tiles x 16 steps; waves 6, 7 have none) -- see model_kernel
constexpr size_t L1PACK_FLOATS = (size_t)WAVES * 56 * 64;
__global__ __launch_bounds__(256) void model_pack_l1_kernel(const float *__restrict__ prm,
                                                            float *__restrict__ l1pack)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= (int)L1PACK_FLOATS) return;
    const int k = e & 3, lane = (e >> 2) & 63, q = (e >> 8) % 14, wave = e / (14 * 256);
    const int slot = q * 4 + k, i = lane & 15, g = lane >> 4;
    float v = 0.0f;
    if (slot < 24) {
        const int jt = wave >> 1, half = wave & 1, j = jt * 16 + i, ww = slot / 8, t = slot % 8;
        v = prm[P_L1W + j * 192 + (t + 8 * g) * 6 + half * 3 + ww];
    } else if (wave < 6) {
        const int ct = (slot - 24) / 16, t = (slot - 24) % 16;
        v = prm[P_L1W + (t + 16 * g) * 192 + (ct * 16 + i) * 6 + wave];
    }
    l1pack[e] = v;
}

template <int MODE, bool GATHER = false>
int model_launch(const float *prm, const float *x, const float *y, const float *dout, float *out,
                 float *gtiles, float *loss_part, long long B, float dscale, int grid,
                 hipStream_t st, float *l1pack = nullptr, ModelGather gs = ModelGather())
{
    // per device: the attribute belongs to the function's image on the current device
    static bool attr_set_dev[64][4] = {};
    int dev = 0;
    WM_HIP(hipGetDevice(&dev));
    bool (&attr_set)[4] = attr_set_dev[dev & 63];
    if (!attr_set[GATHER ? 3 : MODE]) {
        WM_HIP(hipFuncSetAttribute((const void *)model_kernel<MODE, GATHER>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
        attr_set[GATHER ? 3 : MODE] = true;
    }
    if (MODE != 0) {
        hipLaunchKernelGGL(model_pack_l1_kernel, dim3((unsigned)((L1PACK_FLOATS + 255) / 256)),
                           dim3(256), 0, st, prm, l1pack);
        WM_LAUNCH_CHECK("model_pack_l1_kernel");
    }
    hipLaunchKernelGGL((model_kernel<MODE, GATHER>), dim3(grid), dim3(THREADS), LDS_BYTES, st, prm,
                       x, y, dout, out, gtiles, loss_part, B, dscale, (const float *)l1pack, gs);
    WM_LAUNCH_CHECK("model_kernel");
    return WM_OK;
}

}  // namespace

extern "C" int wm_model_param_count(void) { return NPARAM; }

extern "C" size_t wm_model_workspace_bytes(int64_t B)
{
    if (B <= 0 || wm_num_cus() <= 0) return 0;
    const size_t blocks = (size_t)wm_num_cus();
    return wm_align_up(blocks * NTILES * 256 * sizeof(float) + blocks * sizeof(float) + 256, 256) +
           (size_t)GSPLIT * NTILES * 256 * sizeof(double) + L1PACK_FLOATS * sizeof(float);
}

extern "C" int wm_model_fwd_f32(const float *params, const float *x, float *out, int64_t B,
                                void *stream)
{
    WM_REQUIRE(params && x && out, WM_ERR_INVALID, "wm_model_fwd_f32: null pointer");
    WM_REQUIRE(B >= 0, WM_ERR_INVALID, "wm_model_fwd_f32: negative batch");
    if (B == 0) return WM_OK;
    WM_REQUIRE(wm_num_cus() > 0, WM_ERR_NO_DEVICE, "wm_model_fwd_f32: no HIP device");
    return model_launch<0>(params, x, nullptr, nullptr, out, nullptr, nullptr, B, 0.0f,
                           model_grid(B), (hipStream_t)stream);
}

static int model_train_impl(const float *params, const float *x, const float *y, const float *dout,
                            const ModelGather *gs, float *out, float *loss, float *grads,
                            void *workspace, size_t workspace_bytes, int64_t B, int64_t B_global,
                            void *stream);

extern "C" int wm_model_train_f32(const float *params, const float *x, const float *y,
                                  const float *dout, float *out, float *loss, float *grads,
                                  void *workspace, size_t workspace_bytes, int64_t B,
                                  int64_t B_global, void *stream)
{
    WM_REQUIRE(params && x && out && grads, WM_ERR_INVALID, "wm_model_train_f32: null pointer");
    WM_REQUIRE((y != nullptr) != (dout != nullptr), WM_ERR_INVALID,
               "wm_model_train_f32: pass exactly one of y (fused MSE) and dout (given gradient)");
    return model_train_impl(params, x, y, dout, nullptr, out, loss, grads, workspace,
                            workspace_bytes, B, B_global, stream);
}

extern "C" int wm_model_train_gather_f32(const float *params, const float *vol, const int64_t *idx,
                                         int64_t idx0, int T, int Z, int Y, int X, int St, int Sz,
                                         int Sy, int Sx, int time_offset, int d_mid, int d_end,
                                         float *out, float *loss, float *grads, void *workspace,
                                         size_t workspace_bytes, int64_t B, int64_t B_global,
                                         void *stream)
{
    WM_REQUIRE(params && vol && out && grads, WM_ERR_INVALID,
               "wm_model_train_gather_f32: null pointer");
    WM_REQUIRE(T > 0 && Z > 0 && Y > 0 && X > 0 && St > 0 && Sz > 0 && Sy > 0 && Sx > 0 &&
                   Sz <= Z && Sy <= Y && Sx <= X,
               WM_ERR_INVALID, "wm_model_train_gather_f32: sample shape outside the volume");
    WM_REQUIRE(time_offset >= 0 && d_mid >= 0 && d_end >= 0 &&
                   time_offset + St - 1 + (d_end > d_mid ? d_end : d_mid) < T,
               WM_ERR_INVALID, "wm_model_train_gather_f32: time %d + %d steps + offset %d leaves "
               "the %d slices", time_offset, St, d_end > d_mid ? d_end : d_mid, T);
    const long long tot = (long long)St * Sz * Sy * Sx;
    WM_REQUIRE(tot < (1ll << 32) && (idx || idx0 >= 0), WM_ERR_UNSUPPORTED,
               "wm_model_train_gather_f32: sample shape of 2^32 entries or more, or negative index");
    ModelGather gs;
    gs.vol = vol;
    gs.idx = (const long long *)idx;
    gs.idx0 = idx0;
    gs.plane = (unsigned long long)Z * Y * X * 6;
    gs.tot = (unsigned)tot;
    gs.Sx = Sx; gs.Sy = Sy; gs.Sz = Sz; gs.St = St; gs.Y = Y; gs.X = X;
    gs.time_offset = time_offset; gs.d_mid = d_mid; gs.d_end = d_end;
    return model_train_impl(params, nullptr, nullptr, nullptr, &gs, out, loss, grads, workspace,
                            workspace_bytes, B, B_global, stream);
}

static int model_train_impl(const float *params, const float *x, const float *y, const float *dout,
                            const ModelGather *gsp, float *out, float *loss, float *grads,
                            void *workspace, size_t workspace_bytes, int64_t B, int64_t B_global,
                            void *stream)
{
    WM_REQUIRE(B > 0 && B_global >= B, WM_ERR_INVALID, "wm_model_train_f32: bad batch sizes");
    WM_REQUIRE(wm_num_cus() > 0, WM_ERR_NO_DEVICE, "wm_model_train_f32: no HIP device");
    const size_t need = wm_model_workspace_bytes(B);
    WM_REQUIRE(workspace && workspace_bytes >= need, WM_ERR_WORKSPACE,
               "wm_model_train_f32: workspace of %zu bytes needed, %zu given", need,
               workspace_bytes);
    hipStream_t st = (hipStream_t)stream;
    const int grid = model_grid(B);
    float *gtiles = (float *)workspace;
    float *loss_part = gtiles + (size_t)wm_num_cus() * NTILES * 256;
    const float dscale = 1.0f / (6.0f * (float)B_global);
    int rc;
    float *l1pack = (float *)((char *)workspace + need - L1PACK_FLOATS * sizeof(float));
    if (gsp)
        rc = model_launch<1, true>(params, nullptr, nullptr, nullptr, out, gtiles, loss_part, B,
                                   dscale, grid, st, l1pack, *gsp);
    else if (y) rc = model_launch<1>(params, x, y, nullptr, out, gtiles, loss_part, B, dscale, grid, st, l1pack);
    else   rc = model_launch<2>(params, x, nullptr, dout, out, gtiles, loss_part, B, dscale, grid, st, l1pack);
    if (rc) return rc;
    double *tsum = (double *)((char *)l1pack - (size_t)GSPLIT * NTILES * 256 * sizeof(double));
    hipLaunchKernelGGL(model_tile_reduce_kernel, dim3(NTILES, GSPLIT), dim3(256), 0, st, gtiles, grid,
                       tsum);
    WM_LAUNCH_CHECK("model_tile_reduce_kernel");
    hipLaunchKernelGGL(model_grad_finalize_kernel, dim3((NPARAM + 255) / 256), dim3(256), 0, st,
                       tsum, (y || gsp) ? loss_part : (const float *)nullptr, grid, grads,
                       (y || gsp) ? loss : (float *)nullptr, dscale);
    WM_LAUNCH_CHECK("model_grad_finalize_kernel");
    return WM_OK;
}

// ------------------------------------------------------------------------------------------
// Multi-tensor Adam with the step counter on the device (the Conv3D model's optimizer.apply_gradients,
// convolution.py:54 with compile(optimizer="adam"), :185): one launch updates every parameter tensor
// of the model, and because the counter is read and advanced on the device the call can sit in a
// captured HIP graph and be replayed.  A workgroup takes 1024 consecutive elements of one tensor.
struct AdamTensor {
    float *p;
    const float *g;
    float *m;
    float *v;
    long long count;
    long long first_block;
};

namespace {

constexpr int ADAM_MAXT = 16;       // records per launch: they travel as kernel arguments (768 bytes)
struct AdamTable {
    AdamTensor t[ADAM_MAXT];
};

// *step = the number of this update (1 for the first): advanced by adam_step_advance_kernel BEFORE the
// first launch of a step, so that several launches of one step (a layer updated early from its
// gradient hook, the rest at the end) read the same value and the one-thread launch can sit on a side
// stream, off the step's critical path.  (A last-workgroup ticket inside this kernel was tried
// instead of the extra launch: 5600 same-address atomics took the step from 0.41 to 0.60 ms.)
__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamTable tabv, int ntab,
                                                         const int *__restrict__ step, double lr,
                                                         double beta1, double beta2, float eps,
                                                         int keras_eps)
{
    __shared__ float sc[2];
    if (threadIdx.x == 0) {
        const double t = (double)(*step);
        const double bc1 = 1.0 - pow(beta1, t), bc2 = 1.0 - pow(beta2, t);
        sc[0] = keras_eps ? (float)(lr * sqrt(bc2) / bc1) : (float)(lr / bc1);
        sc[1] = (float)sqrt(bc2);
    }
    const AdamTensor *tab = tabv.t;
    int k = 0;
    while (k + 1 < ntab && tab[k + 1].first_block <= (long long)blockIdx.x) ++k;
    const AdamTensor t = tab[k];
    __syncthreads();
    const float one_minus_b1 = (float)(1.0 - beta1), b2 = (float)beta2, one_minus_b2 = (float)(1.0 - beta2);
    const long long base = ((long long)blockIdx.x - t.first_block) * 1024;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long i = base + u * 256 + threadIdx.x;
        if (i < t.count) {
            const float g = t.g[i];
            float pv = t.p[i], mv = t.m[i], vv = t.v[i];
            if (keras_eps) {
                // keras.optimizers.Adam.update_step: m += (g - m)(1 - b1); v += (g^2 - v)(1 - b2);
                // var -= m * alpha / (sqrt(v) + eps), alpha = lr sqrt(1 - b2^t) / (1 - b1^t)
                mv = mv + (g - mv) * one_minus_b1;
                vv = vv + (g * g - vv) * one_minus_b2;
                pv = pv - (mv * sc[0]) / (sqrtf(vv) + eps);
            } else {
                const AdamScalars a{one_minus_b1, b2, one_minus_b2, eps, sc[0], sc[1], 1.0f};
                adam_update(pv, mv, vv, g, a);
            }
            t.p[i] = pv;
            t.m[i] = mv;
            t.v[i] = vv;
        }
    }
}

__global__ void adam_step_advance_kernel(int *step) { if (threadIdx.x == 0) *step += 1; }

}  // namespace

extern "C" int wm_adam_multi_f32(const void *table, int ntensors, int64_t total_blocks, int *step,
                                 double lr, double beta1, double beta2, double eps, int keras_eps,
                                 int advance, void *stream)
{
    WM_REQUIRE(table && step && ntensors > 0 && total_blocks > 0 && total_blocks < (1ll << 31),
               WM_ERR_INVALID, "wm_adam_multi_f32: bad arguments");
    const AdamTensor *rec = (const AdamTensor *)table;
    if (advance) {
        hipLaunchKernelGGL(adam_step_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step);
        WM_LAUNCH_CHECK("adam_step_advance_kernel");
    }
    for (int t0 = 0; t0 < ntensors; t0 += ADAM_MAXT) {          // the records go in as kernel arguments
        const int nt = ntensors - t0 < ADAM_MAXT ? ntensors - t0 : ADAM_MAXT;
        AdamTable tv = {};
        const long long b0 = rec[t0].first_block;
        for (int k = 0; k < nt; ++k) {
            WM_REQUIRE(rec[t0 + k].p && rec[t0 + k].g && rec[t0 + k].m && rec[t0 + k].v &&
                           rec[t0 + k].count > 0, WM_ERR_INVALID,
                       "wm_adam_multi_f32: record %d has a null pointer or no elements", t0 + k);
            tv.t[k] = rec[t0 + k];
            tv.t[k].first_block -= b0;
        }
        const long long b1 = t0 + nt < ntensors ? rec[t0 + nt].first_block : (long long)total_blocks;
        WM_REQUIRE(b1 > b0, WM_ERR_INVALID, "wm_adam_multi_f32: first_block must increase");
        hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)(b1 - b0)), dim3(256), 0,
                           (hipStream_t)stream, tv, nt, (const int *)step, lr, beta1, beta2, (float)eps,
                           keras_eps);
        WM_LAUNCH_CHECK("adam_multi_kernel");
    }
    return WM_OK;
}

extern "C" int wm_adam_step_f32(float *params, const float *grads, float *m, float *v,
                                int64_t count, int step, double lr, double beta1, double beta2,
                                double eps, float grad_scale, void *stream)
{
    WM_REQUIRE(params && grads && m && v, WM_ERR_INVALID, "wm_adam_step_f32: null pointer");
    WM_REQUIRE(count >= 0 && step >= 1, WM_ERR_INVALID, "wm_adam_step_f32: bad count/step");
    if (count == 0) return WM_OK;
    // bias corrections in double, like torch (1 - beta ** step in Python floats): in float32
    // 1 - 0.999f is already off by 1.3e-5 relative, a systematic error in every early update
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const AdamScalars a{(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                        (float)(lr / bc1), (float)sqrt(bc2), grad_scale};
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, params, grads, m, v, (long long)count, a);
    WM_LAUNCH_CHECK("adam_kernel");
    return WM_OK;
}
