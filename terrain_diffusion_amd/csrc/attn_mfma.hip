// MFMA attention for gfx950: softmax(scale * Q K^T) V with generic query / key lengths and head dims 8..160, bf16 operands, fp32 accumulate.
// Replaces the attention einsums of UNetBlock.attn (terrain_diffusion/models/unet_block.py:102-108: per-token unit-RMS q, k, v, logits / sqrt(C))
// and covers the SD-v1.5 shapes of the panorama demo (annotated_infinite_panorama.py:109-134: self-attention 4096 x 4096 / 1024 x 1024 /
// 256 x 256 with head dims 40 / 80 / 160, cross-attention against 77 text tokens).
//
// Two kernels:
//   attn_pack_kernel  reads q, k, v through arbitrary element strides (the U-Net's interleaved NHWC qkv tensor, or plain [B][H][L][D]), applies
//                     the optional per-token unit-RMS normalisation x / (1e-4 + ||x|| / sqrt(D)) (mp_layers.py:9-12, dim=2 of the
//                     (N, heads, C, 3, HW) view), and writes bf16 operands in the shapes the MFMA kernel wants: Qp / Kp [B][H][L][Dp] (d padded to a
//                     multiple of 16 with zeros) and V TRANSPOSED Vt [B][H][Dm][Lkp] (Dm = D rounded up to 32, keys padded to 64) with the 16 keys
//                     of every group stored in the order {0-3, 8-11, 4-7, 12-15} -- see below.
//   attn_mfma_kernel  flash-style forward.  One wave owns 32 queries; a 4-wave workgroup shares 64-key K / Vt tiles staged in LDS.
//     * S^T = K Q^T on v_mfma_f32_32x32x16_bf16 with K as the A operand (rows = keys) and Q as the B operand (columns = queries): a lane then holds
//       ONE query column (lane & 31) and 16 key rows per 32-key block, so the softmax row maximum / sum is a lane-local reduction plus one exchange
//       with lane ^ 32 -- no LDS round trip, no 32-lane butterfly (guide T12).
//     * O^T = V^T P^T with V^T as the A operand (rows = d) and P as the B operand: the B fragment of a 16-key step wants, per lane, 8 consecutive
//       keys of its query -- which are exactly the 8 probabilities the lane already holds for that step, IF the contraction index (keys) is taken
//       in the order the S^T accumulator delivers them (half-wave 0: keys 0-3, 8-11; half-wave 1: keys 4-7, 12-15 of each 16).  A contraction
//       index may be permuted freely as long as both operands agree, so V^T is simply stored with that key order (attn_pack_kernel) and P never
//       moves between lanes.
//     * online softmax in fp32 with exp2; keys beyond Lk are masked to -inf; O is rescaled per tile.  Two roundings differ from the textbook form
//       (round 3, "lean softmax"), both inside the bf16-operand error budget and both exercised by tests/test_gpu_attention.py:
//       (1) scale * log2(e) is folded into Q BEFORE Q is rounded to bf16 (one multiply per score saved): for a scale that is not a power of two
//           this is one extra bf16 rounding of Q, <= 2^-9 relative per element, the same size as the operand rounding itself;
//       (2) the denominator sums the fp32 probabilities while the numerator contracts their bf16 roundings: numerator and denominator are not
//           normalised against identical values any more; the mismatch is the mean of the rounding errors of the weights of one query --
//           unbiased, <= 2^-9 relative and shrinking with the number of keys (measured 1.2-1.5e-3 rel-RMS against a bf16-operand reference
//           on every shape of the test, 4096 keys at d = 40 with scale 0.173 included; bound in the test 4e-3).
//       (3) deferred maximum (round 4, TD_ATTN_THR = 8): the reference point of a query's exponentials follows its running maximum only when
//           some query of the wave has outgrown its own by more than 2^8; in between the probabilities of a tile are <= 2^8 instead of <= 1
//           (fp32 accumulators, same bf16 relative precision, numerator and denominator share the reference point).  It removes the rescale
//           of O -- a dependent, packed-multiply pass in front of the PV MFMAs -- from most tiles: +6 % / +3 % / +8 % at d = 40 / 64 / 128 on
//           the 4096 x 4096 problem, rel-RMS against the fp64 reference 1.55e-3 -> 1.65e-3 (profiles/r04_attention_mfma_utilisation.txt).
// K / V^T rows in LDS are padded to an odd number of 16-byte slots, which makes every 16-lane ds_read_b128 group conflict-free.
// Consecutive MFMAs go to different accumulators (an instruction between two MFMAs on the SAME accumulator costs ~43 cycles, MI355X_MICROARCH.md).
#include "conv_common.h"

namespace td {

struct AttnStrides { long b, h, t, c; };  // element strides of (batch, head, token, channel)

// Round 6, "folded softmax bookkeeping" -- for head dims that are NOT a multiple of 16 (SD-v1.5's 40; 8, 24, ...), whose operands carry padding anyway.  Two of the
// three per-score VALU passes of the online softmax move to the matrix pipe, which idles 70 % of the time at these head dims:
//   * the subtraction of the reference point m rides in ONE PADDING CHANNEL of the Q K^T contraction: K carries 1.0 there, Q carries -m (bf16; m is kept bf16-
//     representable, so numerator and denominator see exactly the same reference) -- the MFMA delivers s - m;
//   * the row sum l rides in ONE PADDING ROW of V^T (1.0 for real keys): that row of O^T accumulates the sum of the bf16-rounded probabilities -- the values the
//     numerator contracts, so the two are consistent by construction (the fp32 sum of round 3's "lean softmax" was not).
// SD's 4096 x 4096 d = 40 problem: 91.4 -> 76.4 us, MFMA busy 28.5 -> 33.1 % (profiles/r06_attention_folded_softmax.txt).  Where the channel / row has to be PAID for
// (d = 64: one more k-step of Q K^T, one more 32-row block of V^T P^T = 22 instead of 16 MFMAs per 64-key tile, wider K / V^T tiles through LDS) the same kernel is
// 6 - 12 % SLOWER at 38 % MFMA busy -- measured, not shipped: a multiple of 16 keeps the plain form.
static inline bool attn_fold(int D) { return (D & 15) != 0; }
static inline int attn_dp(int D) { return (D + 15) / 16 * 16; }
static inline int attn_dm(int D) { return (D + 31) / 32 * 32; }

// grid (ceil(L / 64), H, B) x 3 roles via blockIdx.x ranges is overkill: one launch per operand (which = 0 q, 1 k, 2 v)
template <typename TIN>
__global__ __launch_bounds__(256) void attn_pack_kernel(const TIN* __restrict__ src, AttnStrides st, int L, int D, int Dp, int Dm, int Lkp, int which, int normalize,
                                                        float post_scale, __bf16* __restrict__ dst, int fold) {
    // one wave per token: lanes stride over channels
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, h = blockIdx.y, b = blockIdx.z, H = gridDim.y;
    if (tok >= (which == 2 ? Lkp : L)) return;
    const bool real = tok < L;
    const TIN* row = src + b * st.b + h * st.h + (long)tok * st.t;
    float v[3] = {0.f, 0.f, 0.f};  // D <= 192: up to 3 channels per lane
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int c = lane + 64 * j;
        if (real && c < D) { v[j] = (float)row[(long)c * st.c]; ss += v[j] * v[j]; }
    }
    float inv = 1.f;
    if (normalize) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
        inv = 1.f / (1e-4f + sqrtf(ss) / sqrtf((float)D));
    }
    inv *= post_scale;  // Q carries the softmax scale * log2(e): the flash kernel's logits come out of the MFMA ready for exp2 (one multiply per score saved)
    // fold != 0: channel D of K = 1.0 (Q's starts at 0 = reference point 0), row D of V^T = 1.0 for real keys (see attn_fold)
    if (which < 2) {
        __bf16* out = dst + (((long)b * H + h) * L + tok) * Dp;
#pragma unroll
        for (int j = 0; j < 3; ++j) { const int c = lane + 64 * j; if (c < Dp) out[c] = (fold && which == 1 && c == D) ? (__bf16)1.f : (__bf16)(v[j] * inv); }
    } else {
        const int pk = (tok & ~15) | (tok & 3) | (((tok >> 3) & 1) << 2) | (((tok >> 2) & 1) << 3);  // key order inside a group of 16 (see header)
        __bf16* out = dst + ((long)b * H + h) * Dm * Lkp + pk;
#pragma unroll
        for (int j = 0; j < 3; ++j) { const int c = lane + 64 * j; if (c < Dm) out[(long)c * Lkp] = (fold && c == D && real) ? (__bf16)1.f : (__bf16)(v[j] * inv); }
    }
}

// The U-Net's own case in ONE launch (round 5; three launches of the generic kernel cost 2.7x the attention they fed at batch 64): the qkv conv's
// NHWC output [b][token][head][d][q|k|v] with d = 64 -- a wave reads the 384 contiguous bytes of one (token, head), lane = d, and writes all three
// operands.  Same arithmetic and reduction order as attn_pack_kernel (lane = channel, xor butterfly 32 ... 1): the same bits.
template <typename TIN>
__global__ __launch_bounds__(256) void attn_pack_qkv64_kernel(const TIN* __restrict__ qkv, int L, int Lkp, long tok_stride, float qscale, __bf16* __restrict__ Qp,
                                                              __bf16* __restrict__ Kp, __bf16* __restrict__ Vt) {
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, h = blockIdx.y, b = blockIdx.z, H = gridDim.y;
    if (tok >= Lkp) return;
    const bool real = tok < L;
    float x[3] = {0.f, 0.f, 0.f};
    if (real) {
        const TIN* row = qkv + ((long)b * L + tok) * tok_stride + (long)h * 192 + lane * 3;
#pragma unroll
        for (int w = 0; w < 3; ++w) x[w] = (float)row[w];
    }
    float inv[3];
#pragma unroll
    for (int w = 0; w < 3; ++w) {
        float ss = x[w] * x[w];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
        inv[w] = 1.f / (1e-4f + sqrtf(ss) / sqrtf(64.f));
    }
    inv[0] *= qscale; inv[1] *= 1.f; inv[2] *= 1.f;
    const long bh = (long)b * H + h;
    if (real) {
        Qp[(bh * L + tok) * 64 + lane] = (__bf16)(x[0] * inv[0]);
        Kp[(bh * L + tok) * 64 + lane] = (__bf16)(x[1] * inv[1]);
    }
    const int pk = (tok & ~15) | (tok & 3) | (((tok >> 3) & 1) << 2) | (((tok >> 2) & 1) << 3);  // key order inside a group of 16 (see header)
    Vt[(bh * 64 + lane) * Lkp + pk] = (__bf16)(x[2] * inv[2]);   // padded keys: zeros
}

#ifndef TD_ATTN_ABL
#define TD_ATTN_ABL 0   // ablation hooks of the pipelined loop (tools/r06_attn_ablate.sh; wrong results by design): 1 no barrier, 2 no staging, 8 no S MFMAs, 16 no PV MFMAs
#endif
#ifndef TD_ATTN_THR
#define TD_ATTN_THR 8
#endif
#ifdef TD_ATTN_TRACE   // tools/attn_bench.hip -DTD_ATTN_TRACE: s_memtime stamps between the phases of a tile, summed per wave, dumped through out_b16
#define TD_AT(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tr_[i] += t_ - tl_; tl_ = t_; }
#else
#define TD_AT(i)
#endif

// DP16 = Dp / 16 (k-steps of Q K^T), DM32 = Dm / 32 (row blocks of O^T)
template <int DP16, int DM32, int NW, bool FOLD, bool PIPE = false>
__global__ __launch_bounds__(64 * NW) void attn_mfma_kernel(const __bf16* __restrict__ Qp, const __bf16* __restrict__ Kp, const __bf16* __restrict__ Vt, float* __restrict__ out_f32,
                                                        __bf16* __restrict__ out_b16, AttnStrides ost, int Lq, int Lk, int Lkp, int D) {
    constexpr int Dp = DP16 * 16, Dm = DM32 * 32, TK = 64;
    constexpr int KSL = (Dp / 8) | 1, KPITCH = KSL * 16;      // K rows: odd number of 16-byte slots
    constexpr int VPITCH = (TK / 8 + 1) * 16;                 // Vt rows: 64 keys = 8 slots -> 9
    // K / V^T tiles are DOUBLE-buffered in LDS (round 3): tile t+1 travels global -> registers while tile t-1 is computed and is written to the
    // other buffer at the top of tile t, so a tile costs one workgroup barrier instead of two
    __shared__ __attribute__((aligned(16))) unsigned char s_k[2][TK * KPITCH];
    __shared__ __attribute__((aligned(16))) unsigned char s_v[2][Dm * VPITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z, H = gridDim.y;
    constexpr int NTHR = 64 * NW;
    const int q = blockIdx.x * (32 * NW) + wave * 32 + l31;
    const bool qok = q < Lq;
    // Q fragments (B operand): lane = query column, half-wave = which 8 of the 16 d of a k-step.  Q already carries scale * log2(e).
    u32x4 qf[DP16];
    {
        const __bf16* qrow = Qp + (((long)b * H + h) * Lq + (qok ? q : 0)) * Dp + lh * 8;
#pragma unroll
        for (int ks = 0; ks < DP16; ++ks) qf[ks] = qok ? *(const u32x4*)(qrow + ks * 16) : u32x4{0u, 0u, 0u, 0u};
    }
    f32x16 o[DM32];
#pragma unroll
    for (int d = 0; d < DM32; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -3.0e38f, l_run = 0.f, m_ref = 0.f;   // FOLD: m_ref = the query's reference point (bf16-representable), m_run = its running maximum relative to it
    const __bf16* kbase = Kp + ((long)b * H + h) * Lk * Dp;
    const __bf16* vbase = Vt + ((long)b * H + h) * Dm * Lkp;
    constexpr int KPIECES = TK * (Dp / 8), VPIECES = Dm * (TK / 8);
    constexpr int KIT = (KPIECES + NTHR - 1) / NTHR, VIT = (VPIECES + NTHR - 1) / NTHR;
    u32x4 kreg[KIT], vreg[VIT];
    // Every load is UNCONDITIONAL (piece index and key row clamped instead of predicated): a lane-predicated load sits in an exec-masked branch,
    // and hipcc's wait-count pass then waits for the loads it has just issued (`s_waitcnt vmcnt(1)` right behind the third request in round 3's
    // ISA: the "prefetch" of a tile stalled the wave for a whole L2 round trip, 19 % of a tile's cycles at d = 128).  A clamped K row is a copy of
    // the last key: its logits are masked in the ragged last tile like any key >= Lk; surplus pieces are loaded and never stored.
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int it = 0; it < KIT; ++it) {
            const int e0 = tid + it * NTHR, e = e0 < KPIECES ? e0 : KPIECES - 1, r = e / (Dp / 8), sl = e % (Dp / 8);
            const int row = k0 + r < Lk ? k0 + r : Lk - 1;
            kreg[it] = *(const u32x4*)(kbase + (long)row * Dp + sl * 8);
        }
#pragma unroll
        for (int it = 0; it < VIT; ++it) {
            const int e0 = tid + it * NTHR, e = e0 < VPIECES ? e0 : VPIECES - 1, r = e / (TK / 8), sl = e % (TK / 8);
            vreg[it] = *(const u32x4*)(vbase + (long)r * Lkp + k0 + sl * 8);   // Lkp is a multiple of 64, padding keys are zero
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int it = 0; it < KIT; ++it) {
            const int e = tid + it * NTHR, r = e / (Dp / 8), sl = e % (Dp / 8);
            if (e < KPIECES) *(u32x4*)(s_k[buf] + r * KPITCH + sl * 16) = kreg[it];
        }
#pragma unroll
        for (int it = 0; it < VIT; ++it) {
            const int e = tid + it * NTHR, r = e / (TK / 8), sl = e % (TK / 8);
            if (e < VPIECES) *(u32x4*)(s_v[buf] + r * VPITCH + sl * 16) = vreg[it];
        }
    };
    // tile t+1 travels global -> registers during tile t-1 .. t and is written to the other LDS buffer at the TOP of tile t (every wave has passed the
    // barrier that ended tile t-1, so that buffer's readers are done); the same registers then take the request for tile t+2.  (Round 3 kept
    // two tiles in registers and copied one register set to the other every tile: 10 % of a tile's cycles in the phase trace.)  Loads are
    // unconditional with the tile index clamped; the store of a tile that does not exist lands in a buffer nobody reads again.
    const int nt_ = (Lk + TK - 1) / TK;
    if constexpr (PIPE) {
    // ---- Software-pipelined tile loop (round 6; plain softmax form with an even DP16, or the folded form).  In the loop below a wave's three phases of a tile are serial -- S MFMAs, ~1000 cycles
    // of softmax VALU, PV MFMAs (phase trace at d = 64: 526 + 1092 + 521 of 3021 cycles per tile and wave) -- and only the SIMD's other wave fills the holes.  Here the
    // matrix work of a tile is issued IN THE SHADOW of the vector work of the same wave: the staged tile t is { K(t+1), V^T(t) }; iteration t issues S(t+1) = K(t+1) Q^T
    // into the second score set while the softmax of S(t) (computed during iteration t-1) runs, then O += V^T(t) P(t) step by step behind the probabilities as they
    // appear.  Program order is pinned by sched_barriers: [S slice | max], [S slice | max, exchange, reference point], (rescale), [S slices | exp chunk 0],
    // [PV step 0 | exp chunk 1], [PV 1 | exp 2], [PV 2 | exp 3], [PV 3]; the fragments of a region are read from LDS one region ahead.  Same MFMAs,
    // same operand order, same softmax arithmetic as the unpipelined loop -- the same bits (tools/attn_bench.hip TD_ATTN_DUMP, cmp against TD_ATTN_PIPE=0).
    static_assert(FOLD || DP16 % 2 == 0, "pipelined loop, plain softmax form: even number of k-steps (four equal S slices)");
    // staged tiles travel global -> registers -> LDS; TWO register sets (the loop is unrolled by two, so the sets alternate without copies): a tile is requested two
    // iterations before it is written to LDS (one iteration of lead left the LDS writes waiting for their loads: 8 % of a tile in the phase trace of the unpipelined loop)
    u32x4 kreg2[KIT], vreg2[VIT];
    auto load_k = [&](u32x4 (&kr)[KIT], int krow0) {
#pragma unroll
        for (int it = 0; it < KIT; ++it) {
            const int e0 = tid + it * NTHR, e = e0 < KPIECES ? e0 : KPIECES - 1, r = e / (Dp / 8), sl = e % (Dp / 8);
            const int row = krow0 + r < Lk ? krow0 + r : Lk - 1;
            kr[it] = *(const u32x4*)(kbase + (long)row * Dp + sl * 8);
        }
    };
    auto load_v = [&](u32x4 (&vr)[VIT], int t) {
        const int k0 = (t < nt_ ? t : nt_ - 1) * TK;
#pragma unroll
        for (int it = 0; it < VIT; ++it) {
            const int e0 = tid + it * NTHR, e = e0 < VPIECES ? e0 : VPIECES - 1, r = e / (TK / 8), sl = e % (TK / 8);
            vr[it] = *(const u32x4*)(vbase + (long)r * Lkp + k0 + sl * 8);
        }
    };
    auto store_k = [&](const u32x4 (&kr)[KIT], int bf) {
#pragma unroll
        for (int it = 0; it < KIT; ++it) {
            const int e = tid + it * NTHR, r = e / (Dp / 8), sl = e % (Dp / 8);
            if (KPIECES % NTHR == 0 || e < KPIECES) *(u32x4*)(s_k[bf] + r * KPITCH + sl * 16) = kr[it];
        }
    };
    auto store_v = [&](const u32x4 (&vr)[VIT], int bf) {
#pragma unroll
        for (int it = 0; it < VIT; ++it) {
            const int e = tid + it * NTHR, r = e / (TK / 8), sl = e % (TK / 8);
            if (VPIECES % NTHR == 0 || e < VPIECES) *(u32x4*)(s_v[bf] + r * VPITCH + sl * 16) = vr[it];
        }
    };
    // S MFMA i of a tile: k-step i / 2, key block i % 2 (consecutive MFMAs alternate accumulators), issued in four slices [SB[j], SB[j+1]).  Folded form: the
    // reference channel of Q sits in the LAST k-step (D / 16 = DP16 - 1), so the last two MFMAs form slices 2 and 3 -- issued behind the point where the reference
    // may move and the Q fragment is rewritten: S(t+1) always comes out relative to the reference the softmax of tile t+1 will assume.
    constexpr int MS = DP16 * 2;
    constexpr int SB[5] = {0, FOLD ? (MS - 1) / 2 : MS / 4, FOLD ? MS - 2 : MS / 2, FOLD ? MS - 1 : 3 * MS / 4, MS};
    auto kfrag = [&](const unsigned char* sk_, int i) { return *(const u32x4*)(sk_ + ((i & 1) * 32 + l31) * KPITCH + (i >> 1) * 32 + lh * 16); };
    auto vfrag = [&](const unsigned char* sv_, int st4, int d) { return *(const u32x4*)(sv_ + (d * 32 + l31) * VPITCH + st4 * 32 + lh * 16); };
    f32x16 sa[2], sb[2];
    // prologue: K(0) through the second buffer for S(0); staged tile 0 into the first; staged tile 1 requested
    load_k(kreg, 0); store_k(kreg, 1);
    load_k(kreg, TK); load_v(vreg, 0); store_k(kreg, 0); store_v(vreg, 0);
    load_k(kreg2, 2 * TK); load_v(vreg2, 1);   // staged tile 1 (written to LDS by iteration 0) in the second set
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sa[kb][r] = 0.f;
#pragma unroll
    for (int i = 0; i < MS; ++i) sa[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kfrag(s_k[1], i)), __builtin_bit_cast(bf16x8, qf[i >> 1]), sa[i & 1], 0, 0, 0);
    __syncthreads();   // K(0)'s readers are done before iteration 0 writes staged tile 1 there
    auto body = [&](auto bufc, f32x16 (&sc)[2], f32x16 (&sn)[2], const int t) {
        constexpr bool BUF = decltype(bufc)::value != 0;
        const int k0 = t * TK;
        const unsigned char* sk_ = s_k[BUF];
        const unsigned char* sv_ = s_v[BUF];
        u32x4 kfr[MS];   // K fragments of the next tile's S MFMAs: slice j is read from LDS one region ahead of its MFMAs
#pragma unroll
        for (int i = SB[0]; i < SB[1]; ++i) kfr[i] = kfrag(sk_, i);   // (in front of the LDS writes: the LDS pipe serves in order)
        __builtin_amdgcn_sched_barrier(0);
        // staged tile t+2 = { K(t+3), V^T(t+2) } is requested here and written to LDS near the end of iteration t+1 (into the buffer iteration t+1 does not read):
        // almost two iterations of lead, and the wait in front of the LDS writes comes AFTER this iteration's requests, so hipcc's wait-count pass can tell the two
        // sets apart (with the writes at the top of the loop it waited for everything but one load: the loop header merges the prologue's and the back edge's state)
#if !(TD_ATTN_ABL & 2)
        if constexpr (BUF) { load_k(kreg2, (t + 3) * TK); load_v(vreg2, t + 2); } else { load_k(kreg, (t + 3) * TK); load_v(vreg, t + 2); }
#endif
        if (k0 + TK > Lk) {  // wave-uniform: only the last tile of a ragged key length
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + kb * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
                    if (key >= Lk) sc[kb][r] = -3.0e38f;
                }
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sn[kb][r] = 0.f;
        auto sslice = [&](int j) {   // the MFMAs of slice j; the fragments of slice j + 1
            if (j < 3) {
#pragma unroll
                for (int i = SB[j + 1]; i < SB[j + 2]; ++i) kfr[i] = kfrag(sk_, i);
            }
#pragma unroll
            for (int i = SB[j]; i < SB[j + 1]; ++i) {
#if TD_ATTN_ABL & 8
                asm volatile("" : "+v"(sn[i & 1]) : "v"(kfr[i]));
#else
                sn[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kfr[i]), __builtin_bit_cast(bf16x8, qf[i >> 1]), sn[i & 1], 0, 0, 0);
#endif
            }
        };
        __builtin_amdgcn_sched_barrier(0);
        // region 1: S slice 0 | first half of the maximum
        sslice(0);
        float mt = fmaxf(sc[0][0], sc[0][1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) mt = fmaxf(fmaxf(mt, sc[0][r]), sc[0][r + 1]);
        __builtin_amdgcn_sched_barrier(0);
        // region 2: S slice 1 | second half of the maximum, exchange, reference point (deferred maximum: see the unpipelined loop)
        sslice(1);
#pragma unroll
        for (int r = 0; r < 16; r += 2) mt = fmaxf(fmaxf(mt, sc[1][r]), sc[1][r + 1]);
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        bool moved;
        float m_new = 0.f, alpha = 1.f;
        if constexpr (FOLD) {
            m_run = fmaxf(m_run, mt);   // (FOLD: the running maximum RELATIVE to m_ref; sc already is logit - m_ref)
            moved = __builtin_amdgcn_ballot_w64(mt > (float)TD_ATTN_THR || m_run < -64.f) != 0;
        } else {
            moved = __builtin_amdgcn_ballot_w64(mt > m_run + (float)TD_ATTN_THR) != 0;
            m_new = moved ? fmaxf(m_run, mt) : m_run;
            alpha = moved ? __builtin_amdgcn_exp2f(m_run - m_new) : 1.f;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (moved) {
            if constexpr (FOLD) {
                // as in the unpipelined loop: new reference = the running maximum rounded to bf16, this tile's scores corrected, the factor clamped for a reference
                // that falls in a query's first tile, the reference channel of the Q fragment (last k-step) rewritten -- in front of S slices 2 and 3, which read it
                const float mr_new = (float)(__bf16)(m_ref + m_run), delta = mr_new - m_ref;
                alpha = __builtin_amdgcn_exp2f(fminf(-delta, 64.f));
                m_ref = mr_new; m_run -= delta;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[kb][r] -= delta;
                const unsigned nb = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)(-mr_new));
                const int e16 = D & 15;
                if (lh == (e16 >> 3)) {
                    const int dw = (e16 & 7) >> 1, hi = e16 & 1;
#pragma unroll
                    for (int w_ = 0; w_ < 4; ++w_)
                        if (w_ == dw) qf[DP16 - 1][w_] = hi ? ((qf[DP16 - 1][w_] & 0x0000ffffu) | (nb << 16)) : ((qf[DP16 - 1][w_] & 0xffff0000u) | nb);
                }
            }
#pragma unroll
            for (int d = 0; d < DM32; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        }
        float psum_e = 0.f, psum_o = 0.f;
        bf16x8 pb[4];   // B fragments of the four 16-key steps: 8 probabilities each, already in contraction order
        auto ehalf = [&](int c, int hh) {
            const int kb = c >> 1, jj = c & 1;
#pragma unroll
            for (int e = hh * 4; e < hh * 4 + 4; e += 2) {
                if constexpr (FOLD) {   // neither the subtraction nor the row sum: both on the matrix pipe (attn_fold)
                    pb[c][e] = (__bf16)__builtin_amdgcn_exp2f(sc[kb][jj * 8 + e]);
                    pb[c][e + 1] = (__bf16)__builtin_amdgcn_exp2f(sc[kb][jj * 8 + e + 1]);
                } else {
                    const float p0 = __builtin_amdgcn_exp2f(sc[kb][jj * 8 + e] - m_new);
                    const float p1 = __builtin_amdgcn_exp2f(sc[kb][jj * 8 + e + 1] - m_new);
                    psum_e += p0; psum_o += p1;
                    pb[c][e] = (__bf16)p0; pb[c][e + 1] = (__bf16)p1;
                }
            }
            if constexpr (!FOLD) asm volatile("" : "+v"(psum_e), "+v"(psum_o));   // the running sums stay in this region (without the pin all 32 additions sink behind the tile's last MFMA)
        };
        __builtin_amdgcn_sched_barrier(0);
        // region 3: S slice 2 | first half of exp chunk 0
        sslice(2);
        ehalf(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // region 4: S slice 3 | second half of exp chunk 0
        u32x4 va[DM32], vb[DM32];
#pragma unroll
        for (int d = 0; d < DM32; ++d) va[d] = vfrag(sv_, 0, d);
        sslice(3);
        ehalf(0, 1);
        __builtin_amdgcn_sched_barrier(0);
        // regions 5 .. 8: PV step s (V^T fragments read one region ahead) | exp chunk s + 1
#pragma unroll
        for (int st4 = 0; st4 < 4; ++st4) {
            u32x4 (&vc)[DM32] = (st4 & 1) ? vb : va;
            u32x4 (&vn)[DM32] = (st4 & 1) ? va : vb;
            if (st4 < 3) {
#pragma unroll
                for (int d = 0; d < DM32; ++d) vn[d] = vfrag(sv_, st4 + 1, d);
            }
#pragma unroll
            for (int d = 0; d < DM32; ++d) {
#if TD_ATTN_ABL & 16
                asm volatile("" : "+v"(o[d]) : "v"(vc[d]), "v"(pb[st4]));
#else
                o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vc[d]), pb[st4], o[d], 0, 0, 0);
#endif
            }
            if (st4 < 3) { ehalf(st4 + 1, 0); ehalf(st4 + 1, 1); }
            if (st4 == 2) {   // staged tile t+1 (requested during iteration t-1) -> the other LDS buffer
#if !(TD_ATTN_ABL & 2)
                if constexpr (BUF) { store_k(kreg, 0); store_v(vreg, 0); } else { store_k(kreg2, 1); store_v(vreg2, 1); }
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (!FOLD) {
            l_run = l_run * alpha + (psum_e + psum_o);
            m_run = m_new;
        }
#if !(TD_ATTN_ABL & 1)
        __syncthreads();
#endif
    };
    for (int t = 0; t < nt_; t += 2) {
        body(std::integral_constant<int, 0>{}, sa, sb, t);
        if (t + 1 < nt_) body(std::integral_constant<int, 1>{}, sb, sa, t + 1);
    }
    } else {
    load_tile(0);
    store_tile(0);
    load_tile(nt_ > 1 ? TK : 0);
    __syncthreads();
    int buf = 0;
#ifdef TD_ATTN_TRACE
    unsigned long long tr_[6] = {0, 0, 0, 0, 0, 0}, tl_ = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    for (int k0 = 0; k0 < Lk; k0 += TK, buf ^= 1) {
        TD_AT(5)
        store_tile(buf ^ 1);
        TD_AT(4)
        { const int k2 = k0 + 2 * TK; load_tile(k2 < Lk ? k2 : (nt_ - 1) * TK); }
        const unsigned char* sk_ = s_k[buf];
        const unsigned char* sv_ = s_v[buf];
        TD_AT(0)
        // ---- S^T = K Q^T for two 32-key blocks
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
        // consecutive MFMAs go to DIFFERENT accumulators: anything issued between two MFMAs on the same accumulator (a fragment read, here) costs
        // ~43 cycles (MI355X_MICROARCH.md), between different ones ~6
#pragma unroll
        for (int ks = 0; ks < DP16; ++ks)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const u32x4 kf = *(const u32x4*)(sk_ + (kb * 32 + l31) * KPITCH + ks * 32 + lh * 16);
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[ks]), s[kb], 0, 0, 0);
            }
#ifdef TD_ATTN_TRACE
        asm volatile("s_nop 0" :: "v"(s[0][0]), "v"(s[1][0]));   // the stamp waits for the MFMA results
#endif
        TD_AT(1)
        // ---- online softmax for this lane's query: keys of register r of block kb = k0 + 32 kb + 8 (r / 4) + 4 lh + r % 4.
        // The softmax is what bounds this kernel at small head dims (d = 40: 14 MFMAs = 448 matrix cycles against ~850 VALU cycles per tile
        // in round 2), so it is kept lean: no scaling multiply (folded into Q), the key mask only on the last, ragged tile, three-input
        // maxima, one v_cvt_pk per two probabilities -- and, FOLD (round 6), neither the subtraction of the reference point nor the row sum
        // (both on the matrix pipe: attn_fold).
        if (k0 + TK > Lk) {  // wave-uniform: only the last tile of a ragged key length
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + kb * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
                    if (key >= Lk) s[kb][r] = -3.0e38f;
                }
        }
        float mt = fmaxf(s[0][0], s[0][1]);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = (kb == 0 ? 2 : 0); r < 16; r += 2) mt = fmaxf(fmaxf(mt, s[kb][r]), s[kb][r + 1]);   // v_max3_f32
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        u32x4 pf[4];  // B fragments of the four 16-key steps: 8 probabilities each, already in contraction order
        bool moved;
        float alpha = 1.f;
        if constexpr (FOLD) {
            // s already is (logit - m_ref): the reference point m_ref of this lane's query rides in channel D of its Q fragment.  It moves -- for the whole
            // wave -- when some query's scores outgrew it by 2^THR, or when a query's running maximum lies so far BELOW it that the probabilities would lose
            // their exponent range (fp32 / bf16 keep relative precision down to 2^-126; 2^-64 is the margin).  The new reference is the query's running
            // maximum, rounded to bf16 (it must be exactly what the MFMA subtracts).
            m_run = fmaxf(m_run, mt);   // (FOLD: the running maximum RELATIVE to m_ref)
            moved = __builtin_amdgcn_ballot_w64(mt > (float)TD_ATTN_THR || m_run < -64.f) != 0;
            if (moved) {
                const float m_new = (float)(__bf16)(m_ref + m_run), delta = m_new - m_ref;   // exact: both bf16 values
                // A reference can only fall by more than 2^64 in a query's FIRST tile (its running maximum never decreases, and after any move it sits at the
                // reference): O and the row sum are still exactly 0 there, so the factor is clamped instead of overflowing to inf (0 x inf = NaN); everywhere else
                // |delta| <= 64 + THR and the factor is exact.
                alpha = __builtin_amdgcn_exp2f(fminf(-delta, 64.f));
                m_ref = m_new; m_run -= delta;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[kb][r] -= delta;   // this tile's scores were contracted against the old reference
                // the Q fragment's reference channel: element D % 8 of the lanes whose half holds channels 8 (D / 8 % 2) ... of k-step D / 16
                const unsigned nb = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)(-m_new));
                const int e16 = D & 15, ksm = D >> 4;
                if (lh == (e16 >> 3)) {
                    const int dw = (e16 & 7) >> 1, hi = e16 & 1;
#pragma unroll
                    for (int ks = 0; ks < DP16; ++ks)
                        if (ks == ksm) {
#pragma unroll
                            for (int w_ = 0; w_ < 4; ++w_)
                                if (w_ == dw) qf[ks][w_] = hi ? ((qf[ks][w_] & 0x0000ffffu) | (nb << 16)) : ((qf[ks][w_] & 0xffff0000u) | nb);
                        }
                }
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    bf16x8 pb;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        pb[e] = (__bf16)__builtin_amdgcn_exp2f(s[kb][jj * 8 + e]);
                        pb[e + 1] = (__bf16)__builtin_amdgcn_exp2f(s[kb][jj * 8 + e + 1]);
                    }
                    pf[kb * 2 + jj] = __builtin_bit_cast(u32x4, pb);
                }
        } else {
        // Deferred maximum (TD_ATTN_THR, in log2 units): the reference point m of a query only follows its running maximum when SOME query of
        // the wave has outgrown its own by more than 2^THR -- then O and the denominator of the whole wave are rescaled; otherwise m stays and the
        // probabilities of this tile are at most 2^THR instead of 1 (fp32 accumulators, bf16 relative precision: nothing overflows, numerator
        // and denominator refer to the same m).  THR = 0 is the textbook form (rescale whenever any maximum grew: most tiles of random data).
        moved = __builtin_amdgcn_ballot_w64(mt > m_run + (float)TD_ATTN_THR) != 0;
        const float m_new = moved ? fmaxf(m_run, mt) : m_run;
        alpha = moved ? __builtin_amdgcn_exp2f(m_run - m_new) : 1.f;
        // plain (unpacked) fp32 subtracts and adds: beside the partner wave's MFMAs a v_pk_add_f32 costs ~13 cycles more than a plain VALU op
        // (MI355X_MICROARCH.md); two running sums (even / odd elements) keep the summation order of the packed form
        float psum_e = 0.f, psum_o = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                bf16x8 pb;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const float p0 = __builtin_amdgcn_exp2f(s[kb][jj * 8 + e] - m_new);       // inputs <= 0, flushing tiny results to 0 is fine
                    const float p1 = __builtin_amdgcn_exp2f(s[kb][jj * 8 + e + 1] - m_new);
                    psum_e += p0; psum_o += p1;                                               // fp32 denominator (the numerator rounds to bf16: 2^-9 relative, unbiased)
                    pb[e] = (__bf16)p0; pb[e + 1] = (__bf16)p1;
                }
                pf[kb * 2 + jj] = __builtin_bit_cast(u32x4, pb);
            }
        l_run = l_run * alpha + (psum_e + psum_o);
        m_run = m_new;
        }
#ifdef TD_ATTN_TRACE
        asm volatile("s_nop 0" :: "v"(pf[3]), "v"(l_run));
#endif
        TD_AT(2)
        // ---- O^T = alpha O^T + V^T P^T
        if (moved) {
#pragma unroll
            for (int d = 0; d < DM32; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        }
#pragma unroll
        for (int st4 = 0; st4 < 4; ++st4)
#pragma unroll
            for (int d = 0; d < DM32; ++d) {
                const u32x4 vf = *(const u32x4*)(sv_ + (d * 32 + l31) * VPITCH + st4 * 32 + lh * 16);
                o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, pf[st4]), o[d], 0, 0, 0);
            }
#ifdef TD_ATTN_TRACE
        asm volatile("s_nop 0" :: "v"(o[0][0]), "v"(o[DM32 - 1][0]));
#endif
        TD_AT(3)
        __syncthreads();
    }
#ifdef TD_ATTN_TRACE
    if (lane == 0 && out_b16) {
        unsigned long long* tb = (unsigned long long*)out_b16 + ((((size_t)b * H + h) * gridDim.x + blockIdx.x) * NW + wave) * 8;
        for (int i = 0; i < 6; ++i) tb[i] = tr_[i];
    }
    out_b16 = nullptr;
#endif
    }   // (unpipelined loop)
    float linv;
    if constexpr (FOLD) {
        // row D of O^T is the denominator: register r, lane half lhl of 32-row block D / 32 (row = 8 (r / 4) + 4 lh + r % 4)
        const int rho = D & 31, lhl = (rho >> 2) & 1, rl = (rho >> 3) * 4 + (rho & 3);
        float lsum = 0.f;
#pragma unroll
        for (int d = 0; d < DM32; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) if (d == (D >> 5) && r == rl) lsum = o[d][r];
        const float other = __shfl_xor(lsum, 32);
        linv = 1.f / (lh == lhl ? lsum : other);
        (void)l_run; (void)m_ref;
    } else {
        linv = 1.f / (l_run + __shfl_xor(l_run, 32));
    }
    if (qok) {
#pragma unroll
        for (int d = 0; d < DM32; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = d * 32 + 8 * g + 4 * lh;  // four consecutive channels
                const long off = b * ost.b + h * ost.h + (long)q * ost.t;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c0 + e < D) {
                        const float val = o[d][g * 4 + e] * linv;
                        if (out_f32) out_f32[off + (long)(c0 + e) * ost.c] = val; else out_b16[off + (long)(c0 + e) * ost.c] = (__bf16)val;
                    }
            }
    }
}

// ---- host launchers
template <typename TIN>
static hipError_t attn_pack(const TIN* q, const TIN* k, const TIN* v, AttnStrides sq, AttnStrides sk, AttnStrides sv, int B, int H, int Lq, int Lk, int D, int normalize,
                            float scale, __bf16* Qp, __bf16* Kp, __bf16* Vt, hipStream_t st) {
    const float qscale = scale * 1.4426950408889634f;  // folded into Q (see attn_pack_kernel)
    const int Dp = attn_dp(D), Dm = attn_dm(D), Lkp = (Lk + 63) / 64 * 64, fold = attn_fold(D) ? 1 : 0;
    hipLaunchKernelGGL(attn_pack_kernel<TIN>, dim3((Lq + 3) / 4, H, B), dim3(256), 0, st, q, sq, Lq, D, Dp, Dm, Lkp, 0, normalize, qscale, Qp, fold);
    hipLaunchKernelGGL(attn_pack_kernel<TIN>, dim3((Lk + 3) / 4, H, B), dim3(256), 0, st, k, sk, Lk, D, Dp, Dm, Lkp, 1, normalize, 1.f, Kp, fold);
    hipLaunchKernelGGL(attn_pack_kernel<TIN>, dim3((Lkp + 3) / 4, H, B), dim3(256), 0, st, v, sv, Lk, D, Dp, Dm, Lkp, 2, normalize, 1.f, Vt, fold);
    return hipGetLastError();
}

template <typename TIN>
static hipError_t attn_pack_qkv64(const TIN* qkv, long tok_stride, int B, int H, int L, float scale, __bf16* Qp, __bf16* Kp, __bf16* Vt, hipStream_t st) {
    const int Lkp = (L + 63) / 64 * 64;
    hipLaunchKernelGGL(attn_pack_qkv64_kernel<TIN>, dim3((Lkp + 3) / 4, H, B), dim3(256), 0, st, qkv, L, Lkp, tok_stride, scale * 1.4426950408889634f, Qp, Kp, Vt);
    return hipGetLastError();
}

static hipError_t attn_mfma(const __bf16* Qp, const __bf16* Kp, const __bf16* Vt, float* out_f32, __bf16* out_b16, AttnStrides ost, int B, int H, int Lq, int Lk, int D,
                            hipStream_t st) {
    const int Dp16 = attn_dp(D) / 16, Dm32 = attn_dm(D) / 32, Lkp = (Lk + 63) / 64 * 64;
    const bool fold = attn_fold(D);
    // 8 waves (256 queries) per workgroup when there are enough queries to fill the chip that way: the K / V^T tile is staged once per workgroup
    static const long big_min = getenv("TD_ATTN_BIG_MIN") ? atol(getenv("TD_ATTN_BIG_MIN")) : 256;   // A/B hook (tools/attn_bench.py)
    const bool big = (long)((Lq + 255) / 256) * H * B >= big_min;
    static const int pipe_env = getenv("TD_ATTN_PIPE") ? atoi(getenv("TD_ATTN_PIPE")) : 1;   // A/B hook: 0 = the unpipelined loop everywhere
    // the software-pipelined loop: long key sequences (short ones lose more to the pipeline's fill -- one more staged K tile, two more barriers -- than they gain) on grids
    // that leave a CU one workgroup (with two or more, the other workgroup's waves already fill a wave's serial phases: measured level at 512 workgroups); head dims <= 64 (above, the hot-loop gain
    // is 1 - 4 % and isolated launches of the d = 128 problem LOSE 9 %: 256 VGPRs + scratch; d = 160 spills outright) -- profiles/r06_attention_pipelined_loop.txt
    static const int pipe_min = getenv("TD_ATTN_PIPE_MIN") ? atoi(getenv("TD_ATTN_PIPE_MIN")) : 512;
    static const long pipe_max_wgs = getenv("TD_ATTN_PIPE_MAX_WGS") ? atol(getenv("TD_ATTN_PIPE_MAX_WGS")) : 384;
    const long nwgs = (long)(big ? (Lq + 255) / 256 : (Lq + 127) / 128) * H * B;
    const bool pipe = pipe_env != 0 && Lk >= pipe_min && nwgs <= pipe_max_wgs;
    const dim3 grid(big ? (Lq + 255) / 256 : (Lq + 127) / 128, H, B), blk(big ? 512 : 256);
#define TD_ATTN_CASE(A, M, F) if (Dp16 == A && Dm32 == M && fold == F) {                                                                        \
        if constexpr ((F || A % 2 == 0) && A <= 4) if (pipe && big) { hipLaunchKernelGGL((attn_mfma_kernel<A, M, 8, F, true>), grid, blk, 0, st, Qp, Kp, Vt, out_f32, out_b16, ost, Lq, Lk, Lkp, D); return hipGetLastError(); } \
        if constexpr ((F || A % 2 == 0) && A <= 4) if (pipe) { hipLaunchKernelGGL((attn_mfma_kernel<A, M, 4, F, true>), grid, blk, 0, st, Qp, Kp, Vt, out_f32, out_b16, ost, Lq, Lk, Lkp, D); return hipGetLastError(); } \
        if (big) hipLaunchKernelGGL((attn_mfma_kernel<A, M, 8, F>), grid, blk, 0, st, Qp, Kp, Vt, out_f32, out_b16, ost, Lq, Lk, Lkp, D);        \
        else hipLaunchKernelGGL((attn_mfma_kernel<A, M, 4, F>), grid, blk, 0, st, Qp, Kp, Vt, out_f32, out_b16, ost, Lq, Lk, Lkp, D);            \
        return hipGetLastError(); }
    // (folded form for head dims with a padding channel / row, plain form for multiples of 16)
    TD_ATTN_CASE(1, 1, true) TD_ATTN_CASE(2, 1, true) TD_ATTN_CASE(3, 2, true) TD_ATTN_CASE(4, 2, true) TD_ATTN_CASE(5, 3, true) TD_ATTN_CASE(6, 3, true) TD_ATTN_CASE(7, 4, true) TD_ATTN_CASE(8, 4, true) TD_ATTN_CASE(9, 5, true) TD_ATTN_CASE(10, 5, true)
    TD_ATTN_CASE(1, 1, false) TD_ATTN_CASE(2, 1, false) TD_ATTN_CASE(3, 2, false) TD_ATTN_CASE(4, 2, false) TD_ATTN_CASE(5, 3, false) TD_ATTN_CASE(6, 3, false) TD_ATTN_CASE(7, 4, false) TD_ATTN_CASE(8, 4, false) TD_ATTN_CASE(9, 5, false) TD_ATTN_CASE(10, 5, false)
#undef TD_ATTN_CASE
    return hipErrorInvalidValue;
}

static size_t attn_workspace_elems(int B, int H, int Lq, int Lk, int D, size_t* qn, size_t* kn, size_t* vn) {
    const size_t Dp = attn_dp(D), Dm = attn_dm(D), Lkp = (Lk + 63) / 64 * 64;
    *qn = (size_t)B * H * Lq * Dp; *kn = (size_t)B * H * Lk * Dp; *vn = (size_t)B * H * Dm * Lkp;
    return *qn + *kn + *vn;
}

}  // namespace td
