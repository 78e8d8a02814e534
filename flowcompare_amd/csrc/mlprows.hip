// Row-resident coupling MLP: in_layer -> hidden layers of one reference MLP (models/nets.py:19-30) in ONE launch, with the
// activations of a 128-row band held in REGISTERS as MFMA operands and only the weights streaming through LDS.
//
// Why (round 3): the coupling nets of every shipped configuration are 512 wide (coupling MLP of models/affine_coupling.py:30-46 and
// models/spline_coupling.py:187-210: in_layer K -> 512, then 2 ... 5 hidden layers 512 -> 512 with GELU and the residual pattern of nets.py).
// As separate 128x128-tile GEMM launches each hidden layer DMAs its A row panel into LDS once per column tile (4x) and its weight
// panel once per row tile, 16 KB of LDS-DMA per 48 MFMAs, and every activation crosses HBM between two launches.
//
// Here a workgroup is four waves (ONE per SIMD, up to 512 registers each) that own 32 point rows each for the whole chain:
//   * the product is TRANSPOSED (weights = MFMA A operand, points = B operand, v_mfma_f32_32x32x16_f16, split-fp16 limbs of
//     DESIGN.md section 3): a wave keeps its 32 rows x K <= 512 input as B fragments in registers (32 k-steps x [hi | lo'] x 4 = 256
//     registers; they are loaded once per layer and never pass through LDS);
//   * the layer's weights are ONE linear stream: the pre-tiled fragment-major image PackedLinear::Wf = [N/32][K/16][limb][lane][8 fp16]
//     is consumed front to back (block of 32 output features by block, k-step by k-step), so an LDS-DMA piece is one linear 1 KiB read
//     and LDS is written and read linearly (no swizzle, no bank conflicts); all four waves read the same stage (16 KB = 8 k-steps) and
//     a ring of MR_R stages with MR_D stages in flight runs continuously across blocks and layers;
//   * per 3 MFMAs a wave issues 2 ds_read_b128 and 1/6 DMA piece (the 128x128 LDS-DMA tile: 20 reads and 8 pieces per 24 MFMAs);
//   * a block's epilogue (bias / rank-1 extra-context term in the accumulator init, cross-product fold, residual, exact-erf GELU, limb
//     split) runs one block later, a quarter per stage, under the next block's MFMAs (two accumulator sets); the accumulator layout
//     (feature on the register, point on the lane) becomes the next layer's B fragment with one v_permlane32_swap per register pair;
//   * between layers the activations round-trip through a fragment-major scratch image [32-row band][k-step][limb][lane][16 B] that
//     every lane writes and re-reads ITSELF (same lane, same bytes: no cross-lane or cross-wave hand-off through memory, 1 KiB per
//     store instruction); the last layer writes the row-major limb image [row][K/16][hi 16 | lo' 16] (GemmEpi::A16) that the fused
//     spline / affine output GEMM copies.
// Arithmetic is the split-fp16 GEMM's (same limb split, same k order, same three products per block into main / cross accumulators,
// same GELU, the residual added behind the k loop in both): results are BIT-IDENTICAL to the per-layer launches the engine takes for
// small batches (tests/test_gpu_flow.py::test_every_kernel_variant_agrees..., torch.equal), so a scene's log-probs do not depend on
// the batch it sits in.  Round 4: the last layer's limb split takes run-time scales (MlpRowsParams::out_s1 / out_s2) so that it can
// write the one-accumulator image of the wide fused spline kernel (spline_wide.hip); (1, 2048) reproduces the fixed form's bits.
#include "common.h"
#include "activations.h"
#include "mlprows.h"
#include <type_traits>
#include <cstdio>

namespace fc {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char mr_lds_char;
typedef const __attribute__((address_space(1))) char mr_glb_char;

constexpr int MR_KSTG = 8;                        // k16 steps per stage
constexpr int MR_STAGE = MR_KSTG * 2 * 1024;      // bytes per stage: 8 k-steps x [hi | lo'] x 1 KiB fragment
constexpr int MR_R = 8, MR_D = 6;                 // ring slots, stages in flight
constexpr int MR_PPW = MR_KSTG * 2 / 4;           // DMA pieces per wave and stage
constexpr int MR_NWAIT = (MR_D - 1) * MR_PPW;     // pieces younger than the stage being waited for (lower bound of the wave's VMEM ops issued since)
constexpr int MR_BIAS_OFF = MR_R * MR_STAGE;      // [512] floats: the layer's bias, then [512] floats: layer 0's rank-1 extra-context column
constexpr int MR_LDS = MR_BIAS_OFF + 2 * MR_HID * 4;   // 132 KB

// swaps the upper half of a with the lower half of b (lanes 32..63 of a <-> lanes 0..31 of b); see gemm.hip upper_to_lower for the nops
// NOT volatile: a volatile asm is a barrier for every memory operation in the scheduler's dependence graph -- the stage's weight-fragment
// LDS reads (and with them its MFMAs) could not be placed before a swap at the end of the epilogue quarter, which serialised the quarter's
// VALU work and the stage's MFMAs (first build: 28 % matrix-pipe utilisation)
__device__ __forceinline__ void mr_swap32(unsigned& a, unsigned& b) {
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ unsigned mr_pack(_Float16 a, _Float16 b) {
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}
__device__ __forceinline__ float mr_lo(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu)); }
__device__ __forceinline__ float mr_hi(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)); }

typedef unsigned mr_u4 __attribute__((ext_vector_type(4)));
typedef float mr_f4 __attribute__((ext_vector_type(4)));
// Loads the compiler must not see as loads: beside LDS-DMA traffic hipcc waits vmcnt(0) at the first use of any ordinary global load
// (and at an LDS read that may alias a DMA'd region), which would drain the weight ring once per block.  The results are only used
// behind mr_wait_vm / mr_wait_lgkm, which tie the registers to the counted wait (the asm is the data dependence).
__device__ __forceinline__ mr_u4 mr_gload16(const char* ptr) {
    mr_u4 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(ptr) : "memory");
    return r;
}
template <int N>
__device__ __forceinline__ void mr_wait_vm(mr_u4& a, mr_u4& b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
__device__ __forceinline__ mr_f4 mr_lds_read16(unsigned addr) {
    mr_f4 r;
    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr) : "memory");
    return r;
}
__device__ __forceinline__ void mr_wait_lgkm(mr_f4& a, mr_f4& b, mr_f4& c, mr_f4& d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory");
}

template <int ACT>
__device__ __forceinline__ float mr_act(float v) {
    if constexpr (ACT == FC_ACT_GELU) return fc_gelu(v);
    else if constexpr (ACT == FC_ACT_RELU) return v > 0.f ? v : 0.f;
    else if constexpr (ACT == FC_ACT_ELU) return v > 0.f ? v : expm1f(v);
    else if constexpr (ACT == FC_ACT_LRELU02) return v > 0.f ? v : 0.2f * v;
    else return v;
}

// diagnostic knob 20 = 4 (profiles/micro/mlp_rows_stamps.py): thread 0 of workgroups 0 and 300 stores the shader clock at six points of every
// stage -- 0 stage entry, 1 own DMA pieces landed, 2 barrier passed, 3 next stage's DMA issued, 4 inline-asm section done (bias prefetch,
// residual loads / swaps, fragment flush), 5 the stage's 24 MFMAs + epilogue micro-steps issued -- as stamps[(wg slot * 256 + stage) * 8 + point]
#define MR_STAMP(K_)                                                                                                  \
    if (STAMPS && p.stamps && tid == 0 && (blockIdx.x == 0 || blockIdx.x == 300) && gcur < 256) {                               \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        p.stamps[((blockIdx.x ? 1 : 0) * 256 + (gcur - ((K_) > 2 ? 1 : 0))) * 8 + (K_)] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    }

template <int KS0, int ACT, bool STAMPS = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void mlp_rows_kernel(const MlpRowsParams p) {
    extern __shared__ char smc[];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = blockIdx.x * 128 + wave * 32 + li;                 // this lane's point
    const int row_in = row < p.rows_valid ? row : p.rows_valid - 1;    // pad rows compute a copy of the last valid row (their results are never read): finite values, nothing to mask
    const size_t band = (size_t)blockIdx.x * 4 + wave;                 // 32-row band of this wave
    const size_t band_bytes = (size_t)MR_HID / 16 * 2048;              // one band of a fragment-major activation image: 32 k-steps x 2 KiB

    // ---------------------------------------------------------------- weight stream
    // State of the continuous stream in scalar registers: the next stage's source address, the stages left in its layer, its ring slot.
    // (Reading p.L[layer] per DMA piece was a scalar load + s_waitcnt lgkmcnt(0) -- which also waits for the LDS reads in flight -- in
    // front of every piece: ~70 instead of ~40 cycles per MFMA slot.)
    const char* ws_ptr = reinterpret_cast<const char*>(p.L[0].Wf) + (wave * MR_PPW) * 1024;      // wave-uniform: the lane's 16 bytes are a 32-bit offset (saddr form)
    const unsigned lane16 = lane * 16;
    int ws_left = (p.L[0].ks / MR_KSTG) * MR_NB, ws_layer = 0;
    unsigned ws_slot = 0;                                                // ring slot of the next stage to issue
    auto issue_piece = [&](auto i_tag) __attribute__((always_inline)) {
        constexpr int i = decltype(i_tag)::value;
        char* dst = smc + ws_slot * MR_STAGE + (wave * MR_PPW + i) * 1024;
        __builtin_amdgcn_global_load_lds((mr_glb_char*)(ws_ptr + i * 1024 + lane16), (mr_lds_char*)dst, 16, 0, 0);
        if constexpr (i == MR_PPW - 1) {
            ws_slot = (ws_slot + 1) % MR_R;
            if (--ws_left > 0) ws_ptr += MR_STAGE;
            else if (ws_layer + 1 < p.nlayers) {
                ++ws_layer;
                ws_ptr = reinterpret_cast<const char*>(p.L[ws_layer].Wf) + (wave * MR_PPW) * 1024;
                ws_left = (p.L[ws_layer].ks / MR_KSTG) * MR_NB;
            } else ws_left = 1;
            // (else: the stream has ended; the last stage is re-loaded into slots nobody reads again, so that the counted waits stay uniform)
        }
    };
    auto issue_stage = [&]() __attribute__((always_inline)) {
        issue_piece(std::integral_constant<int, 0>{}); issue_piece(std::integral_constant<int, 1>{});
        issue_piece(std::integral_constant<int, 2>{}); issue_piece(std::integral_constant<int, 3>{});
    };
    static_assert(MR_PPW == 4, "issue_stage / the slot table place four DMA pieces per wave and stage");
    // a layer's bias (and layer 0's extra-context column) live in LDS for the layer: ordinary global loads inside the stream would make
    // hipcc wait vmcnt(0) at their first use and drain the DMA ring with them (scalar loads are not chosen in a kernel that also stores)
    auto stage_bias = [&](int l) __attribute__((always_inline)) {
        const MlpRowsLayer& L = p.L[l];
        if (wave < 2)
            __builtin_amdgcn_global_load_lds((mr_glb_char*)(reinterpret_cast<const char*>(L.bias) + wave * 1024 + lane * 16),
                                             (mr_lds_char*)(smc + MR_BIAS_OFF + wave * 1024), 16, 0, 0);
        else if (l == 0) {
            if (p.rowscal && L.colvec)
                __builtin_amdgcn_global_load_lds((mr_glb_char*)(reinterpret_cast<const char*>(L.colvec) + (wave - 2) * 1024 + lane * 16),
                                                 (mr_lds_char*)(smc + MR_BIAS_OFF + MR_HID * 4 + (wave - 2) * 1024), 16, 0, 0);
            else
                *reinterpret_cast<float4*>(smc + MR_BIAS_OFF + MR_HID * 4 + (wave - 2) * 1024 + lane * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    stage_bias(0);
#pragma unroll
    for (int i = 0; i < MR_D; ++i) issue_stage();
    unsigned gcur = 0;                                                  // global index of the stage being multiplied

    f16x8 xin[MR_HID / 16][2];                                          // this wave's input rows as B fragments: [k-step][hi | lo']
    float omax = 0.f;
    // The first MR_XA k-steps of xin are pinned to AGPRs (an MFMA reads its B operand from either file): the wave needs ~450 registers, the
    // accumulators (AGPR form) take 64 AGPRs, and left alone the allocator keeps xin in VGPRs and SPILLS a part of it to AGPRs -- four
    // v_accvgpr_read per k-step in front of the MFMAs.
    constexpr int MR_XA = 24;
    auto pin_xin = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < MR_HID / 16; ++s)
            if (s < MR_XA && s < ks) { asm volatile("" : "+a"(xin[s][0])); asm volatile("" : "+a"(xin[s][1])); }
    };

    // ---------------------------------------------------------------- layer 0 input: fp32 segments -> limb fragments
    {
        const int k0 = p.segk[0], k1 = k0 + p.segk[1], k2 = k1 + p.segk[2];
        float4 raw[KS0][2];
#pragma unroll
        for (int s = 0; s < KS0; ++s) {
            const int seg = s < k0 ? 0 : (s < k1 ? 1 : 2);
            const int sl = s - (seg == 0 ? 0 : (seg == 1 ? k0 : k1));
            const bool live = s < k2;
            const float* ptr = p.A[live ? seg : 0] + (size_t)row_in * p.lda[live ? seg : 0] + (live ? sl : 0) * 16 + 8 * lh;
            raw[s][0] = *reinterpret_cast<const float4*>(ptr);
            raw[s][1] = *reinterpret_cast<const float4*>(ptr + 4);
            if (!live) { raw[s][0] = make_float4(0.f, 0.f, 0.f, 0.f); raw[s][1] = raw[s][0]; }
        }
#pragma unroll
        for (int s = 0; s < KS0; ++s) {
            const float x[8] = {raw[s][0].x, raw[s][0].y, raw[s][0].z, raw[s][0].w, raw[s][1].x, raw[s][1].y, raw[s][1].z, raw[s][1].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                omax = fmaxf(omax, fabsf(x[e]));
                const _Float16 h = (_Float16)x[e];
                xin[s][0][e] = h;
                xin[s][1][e] = (_Float16)((x[e] - (float)h) * 2048.0f);
            }
        }
    }

    pin_xin(KS0);

    // one block of 32 output features over NSB stages; the previous block's accumulators (pa, pc) are finished meanwhile
    floatx16 accA, corrA, accB, corrB;
    mr_u4 resA_hi, resA_lo, resB_hi, resB_lo;                           // residual fragments of the block being finished: k-steps 2 pb (A), 2 pb + 1 (B)
    mr_f4 bnext[4], cnext[4];                                           // next block's bias (and layer 0's extra-context column) quads, read a stage ahead
    unsigned outq[4][2][2];                                             // finished quads of the block being finished: [quad][hi | lo'][2 words]

    // Control flow inside a block is compile-time only (HAS_RES, the stage count) or branch-free (uniform selects): every runtime branch
    // would split the stage's basic block and keep the scheduler from placing the epilogue's VALU work between the MFMAs.
    auto run_layer = [&](auto ks_tag, auto res_tag, auto first_tag, const int l) __attribute__((always_inline)) {
        constexpr int KS = decltype(ks_tag)::value;
        constexpr bool HAS_RES = decltype(res_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;            // layer 0: rank-1 extra-context term in the accumulator init
        constexpr int NSB = KS / MR_KSTG;
        static_assert(!HAS_RES || NSB == 4, "only the 512 -> 512 hidden layers carry a residual");
        const MlpRowsLayer& L = p.L[l];
        const bool last = l + 1 == p.nlayers;
        const float sp1 = last ? p.out_s1 : 1.0f, sp2 = last ? p.out_s2 : 2048.0f;      // the output's limb split (the last layer may write the one-accumulator form)
        const char* resbase = HAS_RES ? reinterpret_cast<const char*>(p.hbuf[L.res]) + band * band_bytes + lane * 16 : nullptr;
        // output: fragment-major scratch image (k-step stride 2 KiB, limb stride 1 KiB) or, for the last layer, the row-major limb image
        // (k-step stride 64 B, limb stride 32 B)
        char* outbase = last ? reinterpret_cast<char*>(p.out16) + (size_t)row * (MR_HID / 16 * 64) + lh * 16
                             : reinterpret_cast<char*>(p.hbuf[L.out]) + band * band_bytes + lane * 16;
        const int out_ss = last ? 64 : 2048, out_ls = last ? 32 : 1024;
        const float rs = (FIRST && p.rowscal && L.colvec) ? p.rowscal[row_in] : 0.f;     // (without the term the column in LDS is zero)
        const unsigned bias_lds = (unsigned)(uintptr_t)(mr_lds_char*)smc + (unsigned)(MR_BIAS_OFF + 16 * lh);      // LDS byte address of this lane half's first bias quad

        // bias quads of block nb -> bnext (cnext): issued a stage before the block starts, waited for in init_acc
        auto bias_fetch = [&](int nb) __attribute__((always_inline)) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bnext[g] = mr_lds_read16(bias_lds + (unsigned)(nb * 32 + 8 * g) * 4u);      // (two addresses per wave: LDS broadcast)
                if constexpr (FIRST) cnext[g] = mr_lds_read16(bias_lds + (unsigned)(MR_HID + nb * 32 + 8 * g) * 4u);
            }
        };
        // the block's accumulator start values (bias, + layer 0's rank-1 extra-context term): they enter as the C operand of the block's first
        // MFMA (no moves into the accumulator registers); the cross-product accumulator starts from the constant 0
        auto init_vec = [&]() __attribute__((always_inline)) -> floatx16 {
            mr_wait_lgkm(bnext[0], bnext[1], bnext[2], bnext[3]);
            if constexpr (FIRST) mr_wait_lgkm(cnext[0], cnext[1], cnext[2], cnext[3]);
            floatx16 b;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    b[4 * g + t] = bnext[g][t];
                    if constexpr (FIRST) b[4 * g + t] += rs * cnext[g][t];
                }
            return b;
        };

        // Quarter g of a block's epilogue (accumulators pa / pc): features 32 pb + 8 g + 4 lh + 0..3 of this lane's point -> outq[g], cut into 24
        // micro-steps of ~6 VALU instructions: one micro-step is placed behind each of a stage's 24 MFMAs (below).  A micro-step advances TWO
        // values (step m: values 2 (m / 12), 2 (m / 12) + 1, half-step m % 12 of 12) -- a single value's chain is all dependent instructions,
        // which issue at ~7 cycles instead of 4 with one wave per SIMD, and a slot of six of them outlasts its MFMA.  The arithmetic is
        // fc_gelu's and the split-fp16 GEMM epilogue's, operation for operation.
        float ev[2][4], eu[2][4], eg[2][4];            // [quarter & 1]: a stage of a 16- or 24-k-step layer runs two quarters side by side
        auto epi_step = [&](const floatx16& pa, const floatx16& pc, auto g_tag, auto m_tag) __attribute__((always_inline)) {
            constexpr int g = decltype(g_tag)::value, m = decltype(m_tag)::value, pr = m / 12, hs = m % 12, Q = g & 1;
            static_assert(ACT == FC_ACT_GELU, "the interleaved epilogue is written for the exact-erf GELU of the shipped configurations");
            constexpr float C[12] = {7.953413483363647e-10f, -3.609737220244824e-08f, 7.156736501201522e-07f, -8.034509846766014e-06f, 5.399647488957271e-05f, -0.0001862359931692481f, -0.0002285758382640779f, 0.00727660721167922f, -0.05270823836326599f, -0.4591129422187805f, -1.151120901107788f, -0.9999995827674866f};      // fc_gelu's polynomial in w = min(|v|, 7.354), highest degree first
            // (MR_PIN: an empty volatile asm over a step's results.  The steps are pure arithmetic on registers: nothing else ties them to their
            // slot, and instruction selection otherwise emits all 24 of them in front of the stage's first MFMA.)
#define MR_PIN(X_) asm volatile("" : "+v"(X_))
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int t = 2 * pr + tt;
                if constexpr (hs == 0) {
                    float v = fmaf(pc[4 * g + t], 1.0f / 2048.0f, pa[4 * g + t]);      // (= pa + pc / 2048 rounded once: the scaling is exact)
                    if constexpr (HAS_RES) {
                        // residual quad in accumulator order (after the reverse lane swap done when the fragment arrived)
                        const mr_u4& rh = g < 2 ? resA_hi : resB_hi;
                        const mr_u4& rl = g < 2 ? resA_lo : resB_lo;
                        const unsigned hw = (g & 1) ? (pr == 0 ? rh.z : rh.w) : (pr == 0 ? rh.x : rh.y);
                        const unsigned lw = (g & 1) ? (pr == 0 ? rl.z : rl.w) : (pr == 0 ? rl.x : rl.y);
                        v += tt ? limb_join<1>(hw, lw) : limb_join<0>(hw, lw);             // (one v_fma_mix_f32, activations.h)
                    }
                    MR_PIN(v);
                    ev[Q][t] = v;
                } else if constexpr (hs == 1) {
                    float u = fminf(fabsf(ev[Q][t]), 7.353910524340095f);
                    float gp = fmaf(C[0], u, C[1]);
                    MR_PIN(u); MR_PIN(gp);
                    eu[Q][t] = u; eg[Q][t] = gp;
                } else if constexpr (hs >= 2 && hs <= 4) {                      // coefficients 2..4, 5..7, 8..10
                    float gp = eg[Q][t];
#pragma unroll
                    for (int c = 0; c < 3; ++c) gp = fmaf(gp, eu[Q][t], C[3 * hs - 4 + c]);
                    MR_PIN(gp);
                    eg[Q][t] = gp;
                } else if constexpr (hs == 5) {
                    float gp = fmaf(eg[Q][t], eu[Q][t], C[11]);
                    MR_PIN(gp);
                    eg[Q][t] = gp;
                } else if constexpr (hs == 6) {
                    float e = __builtin_amdgcn_exp2f(eg[Q][t]);                // erfc(u) / 2
                    MR_PIN(e);
                    eg[Q][t] = e;
                } else if constexpr (hs == 7) {
                    // fc_gelu's fma(-|v|, e, max(v, 0)); the maximum as asm: behind MR_PIN the compiler no longer knows that v is canonical and
                    // puts a v_max v, v, v in front of fmaxf
                    float m;
                    asm("v_max_f32 %0, 0, %1" : "=v"(m) : "v"(ev[Q][t]));
                    float r = fmaf(-fabsf(ev[Q][t]), eg[Q][t], m);
                    MR_PIN(r);
                    ev[Q][t] = r;
                } else if constexpr (hs == 8) {
                    if (tt == 1) {
                        // the pair's limb words in 6 instructions (activations.h limb_split2s: run-time scales, (1, 2048) = limb_split2's bits)
                        unsigned wh2, wl2;
                        limb_split2s(ev[Q][2 * pr], ev[Q][2 * pr + 1], sp1, sp2, wh2, wl2);
                        MR_PIN(wh2); MR_PIN(wl2);
                        outq[g][0][pr] = wh2;
                        outq[g][1][pr] = wl2;
                    }
                } else if constexpr (hs == 9) {
                    if (tt == 1) {
                        // running maximum of |output| for the range flag (pad rows replicate the last valid row: no masking)
                        float om = fmaxf(omax, fmaxf(fabsf(ev[Q][2 * pr]), fabsf(ev[Q][2 * pr + 1])));
                        MR_PIN(om);
                        omax = om;
                    }
                }
            }
#undef MR_PIN
        };
        auto quarter = [&](const floatx16& pa, const floatx16& pc, auto g_tag) __attribute__((always_inline)) {      // (a whole quarter at once: the layer's last block)
            auto go = [&](auto... ms) __attribute__((always_inline)) { (epi_step(pa, pc, g_tag, ms), ...); };
            go(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{},
               std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{}, std::integral_constant<int, 6>{}, std::integral_constant<int, 7>{},
               std::integral_constant<int, 8>{}, std::integral_constant<int, 9>{}, std::integral_constant<int, 10>{}, std::integral_constant<int, 11>{},
               std::integral_constant<int, 12>{}, std::integral_constant<int, 13>{}, std::integral_constant<int, 14>{}, std::integral_constant<int, 15>{},
               std::integral_constant<int, 16>{}, std::integral_constant<int, 17>{}, std::integral_constant<int, 18>{}, std::integral_constant<int, 19>{},
               std::integral_constant<int, 20>{}, std::integral_constant<int, 21>{}, std::integral_constant<int, 22>{}, std::integral_constant<int, 23>{});
        };
        // quads 2 h, 2 h + 1 of block fb are k-step 2 fb + h of the next layer's input: accumulator order -> fragment order (one lane-half swap per
        // register pair), then two 16-byte stores.  Inline asm: runs at a stage's START, outside the stage's scheduling region (see mr_swap32)
        auto flush_pair = [&](int fb, auto h_tag) __attribute__((always_inline)) {
            constexpr int h = decltype(h_tag)::value, g = 2 * h + 1;
#pragma unroll
            for (int limb = 0; limb < 2; ++limb) {
                mr_swap32(outq[g - 1][limb][0], outq[g][limb][0]);
                mr_swap32(outq[g - 1][limb][1], outq[g][limb][1]);
                const uint4 frag = make_uint4(outq[g - 1][limb][0], outq[g - 1][limb][1], outq[g][limb][0], outq[g][limb][1]);
                *reinterpret_cast<uint4*>(outbase + (2 * fb + h) * out_ss + limb * out_ls) = frag;
            }
        };
        auto flush_limb = [&](int fb, auto h_tag, auto limb_tag) __attribute__((always_inline)) {
            constexpr int h = decltype(h_tag)::value, g = 2 * h + 1, limb = decltype(limb_tag)::value;
            mr_swap32(outq[g - 1][limb][0], outq[g][limb][0]);
            mr_swap32(outq[g - 1][limb][1], outq[g][limb][1]);
            const uint4 frag = make_uint4(outq[g - 1][limb][0], outq[g - 1][limb][1], outq[g][limb][0], outq[g][limb][1]);
            *reinterpret_cast<uint4*>(outbase + (2 * fb + h) * out_ss + limb * out_ls) = frag;
        };
        // residual fragments of block rb, half h (k-step 2 rb + h): issue the loads; finish (wait + reverse swap) before use
        auto res_load = [&](int rb, int h, mr_u4& rh, mr_u4& rl) __attribute__((always_inline)) {
            const char* src = resbase + (size_t)((2 * rb + h) * 2) * 1024;
            rh = mr_gload16(src);
            rl = mr_gload16(src + 1024);
        };
        auto res_fix = [&](mr_u4& rh, mr_u4& rl) __attribute__((always_inline)) {
            unsigned a0 = rh.x, a1 = rh.y, a2 = rh.z, a3 = rh.w, b0 = rl.x, b1 = rl.y, b2 = rl.z, b3 = rl.w;
            mr_swap32(a0, a2); mr_swap32(a1, a3);
            mr_swap32(b0, b2); mr_swap32(b1, b3);
            rh.x = a0; rh.y = a1; rh.z = a2; rh.w = a3; rl.x = b0; rl.y = b1; rl.z = b2; rl.w = b3;
        };

        // The epilogue of block nb - 1 runs under block nb.  Block 0 has no predecessor: it "finishes" block 0 itself from zeroed
        // accumulators -- the fragments it stores at block 0's place are overwritten by the real ones one block later (same lane, same
        // address, program order) -- so that the block body has no runtime branch.
        auto block = [&](floatx16& acc, floatx16& corr, const floatx16& pa, const floatx16& pc, const int nb) __attribute__((always_inline)) {
            const int pb = nb > 0 ? nb - 1 : 0;
            const floatx16 binit = init_vec();
            const floatx16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            // A stage = 8 k-steps = 24 MFMAs.  One wave per SIMD: nothing else fills the matrix pipe while this wave issues anything else, and
            // left alone the scheduler emits a stage's VALU / memory instructions first and its MFMAs after them (first build: 28 % of the matrix
            // peak; sched_group_barrier pipelines were honoured in one stage out of four).  So the order is pinned by hand: the stage is 24
            // SLOTS, each one MFMA followed by what runs in its 32-cycle shadow -- one micro-step of the previous block's epilogue (a quarter
            // per stage when a block has 4 stages, two per stage with 2, 2 + 1 + 1 with 3), and in fixed slots the next k-step's weight
            // fragment reads, the MR_PPW DMA pieces of the stage MR_D ahead, the bias prefetch, the residual loads / waits / lane swaps and
            // the lane swaps + stores of finished fragment pairs -- and a sched_barrier fences every slot.  Only the stage's wait + barrier
            // stand outside.
            auto stage = [&](auto q_tag) __attribute__((always_inline)) {
                constexpr int q = decltype(q_tag)::value;
                MR_STAMP(0)
                // ONE wait + barrier per PAIR of stages (round 3: the wave stood ~150 cycles at the wait and ~160 at the barrier of every
                // 24-MFMA stage): behind the barrier of an even stage everybody's pieces of that stage AND the next have landed, and everybody
                // is done with the stage before -- the slot the pieces issued during the two stages go to (stage g + MR_D lands where
                // stage g - 2 / g - 1 sat).  Layers are an even number of stages long, so a layer starts on an even stage.
                if ((gcur & 1u) == 0u) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MR_NWAIT - MR_PPW) : "memory");       // this wave's pieces of stages gcur, gcur + 1 have landed
                    MR_STAMP(1)
                    __builtin_amdgcn_s_barrier();
                }
                MR_STAMP(2)
                const char* sb = smc + (gcur % MR_R) * MR_STAGE + lane * 16;
                ++gcur;
                __builtin_amdgcn_sched_barrier(0);
                constexpr int QA = NSB == 4 ? q : (NSB == 2 ? 2 * q : (q == 0 ? 0 : q + 1));      // first quarter of this stage
                constexpr int QN = NSB == 4 ? 1 : (NSB == 2 ? 2 : (q == 0 ? 2 : 1));              // quarters in this stage
                constexpr int FLUSH_A = NSB == 4 ? 2 : 1;                                         // stage behind the one that finishes quads 0, 1
                f16x8 wh[2], wl[2];
                wh[0] = *reinterpret_cast<const f16x8*>(sb);
                wl[0] = *reinterpret_cast<const f16x8*>(sb + 1024);
                auto slot = [&](auto i_tag) __attribute__((always_inline)) {
                    constexpr int i = decltype(i_tag)::value;                                   // MFMA index inside the stage, 0..23
                    if constexpr (i % 6 == 1) issue_piece(std::integral_constant<int, i / 6>{});
                    if constexpr (i == 2 && q == NSB - 1) bias_fetch(nb + 1 < MR_NB ? nb + 1 : nb);
                    if constexpr (NSB == 4 && HAS_RES) {
                        // residual halves: loaded two stages ahead of their use, waited for and un-swapped in the last slot of the stage before it
                        // (at least 6 VMEM instructions -- DMA pieces, fragment stores -- are issued in between: the counted wait is a lower bound)
                        if constexpr (i == 3 && q == 0) res_load(pb, 1, resB_hi, resB_lo);
                        if constexpr (i == 3 && q == 2) res_load(nb, 0, resA_hi, resA_lo);
                        if constexpr (i == 23 && q == 1) { mr_wait_vm<6>(resB_hi, resB_lo); res_fix(resB_hi, resB_lo); }
                        if constexpr (i == 23 && q == 3) { mr_wait_vm<6>(resA_hi, resA_lo); res_fix(resA_hi, resA_lo); }
                    }
                    if constexpr (q == 0 && (i == 4 || i == 5))                                   // quads 2, 3 of the block finished under the previous block
                        flush_limb(nb > 1 ? nb - 2 : 0, std::integral_constant<int, 1>{}, std::integral_constant<int, i - 4>{});
                    if constexpr (q == FLUSH_A && (i == 4 || i == 5))                             // quads 0, 1 of the block being finished
                        flush_limb(pb, std::integral_constant<int, 0>{}, std::integral_constant<int, i - 4>{});
                    epi_step(pa, pc, std::integral_constant<int, QA>{}, i_tag);
                    if constexpr (QN == 2) epi_step(pa, pc, std::integral_constant<int, QA + 1>{}, i_tag);
                    __builtin_amdgcn_sched_barrier(0);
                };
                auto kstep = [&](auto j_tag) __attribute__((always_inline)) {
                    constexpr int j = decltype(j_tag)::value, cur = j & 1, nxt = cur ^ 1, sx = q * MR_KSTG + j;
                    if constexpr (sx == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[cur], xin[sx][0], binit, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[cur], xin[sx][0], acc, 0, 0, 0);      // hi * hi
                    if constexpr (j + 1 < MR_KSTG) {
                        wh[nxt] = *reinterpret_cast<const f16x8*>(sb + (2 * j + 2) * 1024);
                        wl[nxt] = *reinterpret_cast<const f16x8*>(sb + (2 * j + 3) * 1024);
                    }
                    slot(std::integral_constant<int, 3 * j>{});
                    if constexpr (sx == 0) corr = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[cur], xin[sx][0], zero16, 0, 0, 0);
                    else corr = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[cur], xin[sx][0], corr, 0, 0, 0);    // lo' * hi
                    slot(std::integral_constant<int, 3 * j + 1>{});
                    corr = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[cur], xin[sx][1], corr, 0, 0, 0);          // hi * lo'
                    slot(std::integral_constant<int, 3 * j + 2>{});
                };
                kstep(std::integral_constant<int, 0>{}); kstep(std::integral_constant<int, 1>{}); kstep(std::integral_constant<int, 2>{});
                kstep(std::integral_constant<int, 3>{}); kstep(std::integral_constant<int, 4>{}); kstep(std::integral_constant<int, 5>{});
                kstep(std::integral_constant<int, 6>{}); kstep(std::integral_constant<int, 7>{});
                MR_STAMP(5)
            };
            stage(std::integral_constant<int, 0>{});
            stage(std::integral_constant<int, 1>{});
            if constexpr (NSB > 2) stage(std::integral_constant<int, 2>{});
            if constexpr (NSB > 3) stage(std::integral_constant<int, 3>{});
        };

#pragma unroll
        for (int r = 0; r < 16; ++r) { accB[r] = 0.f; corrB[r] = 0.f; }
        if constexpr (HAS_RES) { res_load(0, 0, resA_hi, resA_lo); mr_wait_vm<0>(resA_hi, resA_lo); }      // (block 0's stand-in epilogue reads it; any finite bytes do)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");            // the layer's bias has landed (and this wave's input fragments)
        __builtin_amdgcn_s_barrier();
        bias_fetch(0);
        for (int nb = 0; nb < MR_NB; nb += 2) {
            block(accA, corrA, accB, corrB, nb);
            block(accB, corrB, accA, corrA, nb + 1);
        }
        // the last block's epilogue stands alone (nothing of this layer left to multiply under it)
        if constexpr (HAS_RES) {
            res_load(MR_NB - 1, 0, resA_hi, resA_lo);
            res_load(MR_NB - 1, 1, resB_hi, resB_lo);
            mr_wait_vm<0>(resA_hi, resA_lo);
            mr_wait_vm<0>(resB_hi, resB_lo);
            res_fix(resA_hi, resA_lo);
            res_fix(resB_hi, resB_lo);
        }
        flush_pair(MR_NB - 2, std::integral_constant<int, 1>{});
        quarter(accB, corrB, std::integral_constant<int, 0>{});
        quarter(accB, corrB, std::integral_constant<int, 1>{});
        flush_pair(MR_NB - 1, std::integral_constant<int, 0>{});
        quarter(accB, corrB, std::integral_constant<int, 2>{});
        quarter(accB, corrB, std::integral_constant<int, 3>{});
        flush_pair(MR_NB - 1, std::integral_constant<int, 1>{});
    };

    run_layer(std::integral_constant<int, KS0>{}, std::false_type{}, std::true_type{}, 0);
    for (int l = 1; l < p.nlayers; ++l) {
        // this layer's input = the image this wave just finished writing (each lane re-reads its own bytes)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                        // every wave is done with the previous layer's bias (its last read sits behind that layer's last stage barrier)
        stage_bias(l);
        const char* src = reinterpret_cast<const char*>(p.hbuf[p.L[l].in]) + band * band_bytes + lane * 16;
#pragma unroll
        for (int s = 0; s < MR_HID / 16; ++s) {
            xin[s][0] = *reinterpret_cast<const f16x8*>(src + (size_t)(2 * s) * 1024);
            xin[s][1] = *reinterpret_cast<const f16x8*>(src + (size_t)(2 * s + 1) * 1024);
        }
        pin_xin(MR_HID / 16);
        if (p.L[l].res >= 0) run_layer(std::integral_constant<int, MR_HID / 16>{}, std::true_type{}, std::false_type{}, l);
        else run_layer(std::integral_constant<int, MR_HID / 16>{}, std::false_type{}, std::false_type{}, l);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // drain the stream's tail re-loads before the LDS is released
    if (!(omax < p.range_limit) && p.ovf) atomicOr(p.ovf, 1);           // (also on a NaN; 65504 / the last layer's pre-scale: conservative for the layers before it)
}

// ---------------------------------------------------------------- fragment-major weight image
// Wf[nb][s][limb][lane = (li, lh)][8] = W2[32 nb + li][s][limb][8 lh .. 8 lh + 7], zero for s >= K_pad / 16 (layer 0 is padded to a whole stage)
__global__ __launch_bounds__(256) void mr_frag_image_kernel(const unsigned short* __restrict__ W2, int ks_src, int ks_dst, uint4* __restrict__ Wf, long n16) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    const int lane = (int)(i & 63);
    const long f = i >> 6;                       // (nb * ks_dst + s) * 2 + limb
    const int limb = (int)(f & 1);
    const long bs = f >> 1;
    const int s = (int)(bs % ks_dst);
    const long nb = bs / ks_dst;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (s < ks_src) v = *reinterpret_cast<const uint4*>(W2 + (((size_t)(nb * 32 + (lane & 31)) * ks_src + s) * 2 + limb) * 16 + 8 * (lane >> 5));
    Wf[i] = v;
}

int mlp_rows_ks_pad(int K_pad) {
    const int ks = K_pad / 16;
    return ks <= 16 ? 16 : (ks <= 24 ? 24 : 32);
}
size_t mlp_rows_image_bytes(int K_pad) { return (size_t)MR_NB * mlp_rows_ks_pad(K_pad) * 2048; }

void launch_mlp_rows_image(const PackedLinear& L, unsigned short* Wf, hipStream_t s) {
    if (!L.W2 || L.N_pad != MR_HID || L.K_pad % 16 != 0 || L.K_pad > MR_HID || L.n_alloc < MR_HID)
        throw Error(FC_ERR_INVALID, "launch_mlp_rows_image: layer is not a K <= 512 -> 512 layer with an fp16 limb image");
    const int ksd = mlp_rows_ks_pad(L.K_pad);
    const long n16 = (long)MR_NB * ksd * 2 * 64;
    mr_frag_image_kernel<<<dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s>>>(L.W2, L.K_pad / 16, ksd, reinterpret_cast<uint4*>(Wf), n16);
    FC_HIP(hipGetLastError());
}

// row-major limb image [rows][width/16][hi 16 | lo' 16] -> fp32 (fc_op_mlp_hidden_f32: unit tests read the chain's output through it)
__global__ __launch_bounds__(256) void mr_limb_decode_kernel(const unsigned short* __restrict__ img, float* __restrict__ out, int ldo, long n, int width) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long r = i / width;
    const int c = (int)(i - r * width);
    const unsigned short* q = img + ((size_t)r * (width / 16) + (c >> 4)) * 32 + (c & 15);
    out[(size_t)r * ldo + c] = (float)__builtin_bit_cast(_Float16, q[0]) + (float)__builtin_bit_cast(_Float16, q[16]) * (1.0f / 2048.0f);
}
void launch_limb_decode(const unsigned short* img, float* out, int ldo, int rows, int width, hipStream_t s) {
    const long n = (long)rows * width;
    if (n <= 0) return;
    mr_limb_decode_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(img, out, ldo, n, width);
    FC_HIP(hipGetLastError());
}

extern int g_gemm_stamp;
unsigned long long* gemm_stamp_buffer(size_t n);
int g_mlp_rows = 1;          // knob 23: 1 = row-resident coupling MLP chain where it fills the chip (shipped), 2 = at any size, 0 = one GEMM launch per layer

template <int KS0, int ACT, bool STAMPS = false>
static void mr_launch(const MlpRowsParams& p, int rows_alloc, double flops, hipStream_t s) {
    auto kern = mlp_rows_kernel<KS0, ACT, STAMPS>;
    static PerDeviceOnce attr_once;
    attr_once.run([&](int) { FC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, MR_LDS)); return 0; });
    char name[96];
    snprintf(name, sizeof name, "void fc::mlp_rows_kernel<%d, %d, %s>(fc::MlpRowsParams)", KS0, ACT, STAMPS ? "true" : "false");      // (the name rocprofv3 prints)
    ProfScope ps(name, flops, 0.0, s);
    hipLaunchKernelGGL(kern, dim3(rows_alloc / 128), dim3(256), MR_LDS, s, p);
    FC_HIP(hipGetLastError());
}

bool mlp_rows_fills_the_chip(int rows_alloc) {
    static PerDeviceOnce cus_once;
    const int cus = cus_once.run([](int dev) {
        int n = 0;
        FC_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        return n;
    });
    return g_mlp_rows == 2 || rows_alloc / 128 >= (3 * cus) / 4;      // (knob 23 = 2: the chain at any size -- tests)
}

bool mlp_rows_eligible(const PackedLinear& in, const std::vector<PackedLinear>& mid, int act) {
    if (!g_mlp_rows || act != FC_ACT_GELU) return false;
    if (!in.Wf || in.N_pad != MR_HID || in.K_pad > MR_HID || in.nseg < 1 || in.nseg > 3 || !in.bias) return false;
    for (int i = 0; i < in.nseg; ++i) if (in.seg_k[i] % 16 != 0) return false;
    if (mid.empty() || (int)mid.size() + 1 > MR_MAXL) return false;
    for (const PackedLinear& L : mid) if (!L.Wf || L.N_pad != MR_HID || L.K_pad != MR_HID || L.nseg != 1 || !L.bias) return false;
    return true;
}

void launch_mlp_rows(const PackedLinear& in, const std::vector<PackedLinear>& mid, const ASeg* segs, const float* rowscal, int act,
                     float* const h[3], unsigned short* out16, int rows_alloc, int rows_valid, hipStream_t s, float out16_scale) {
    if (!mlp_rows_eligible(in, mid, act)) throw Error(FC_ERR_UNSUPPORTED, "launch_mlp_rows: shapes outside the row-resident chain");
    if (rows_alloc % 128 != 0 || !out16) throw Error(FC_ERR_INVALID, "launch_mlp_rows: rows must be padded to 128 and a limb-image output given");
    int* flag = gemm_fp16_flag();
    if (!flag) throw Error(FC_ERR_INVALID, "launch_mlp_rows: needs an open split-fp16 guard scope");
    MlpRowsParams p{};
    for (int i = 0; i < 3; ++i) {
        p.A[i] = i < in.nseg ? segs[i].ptr : segs[0].ptr;
        p.lda[i] = i < in.nseg ? segs[i].lda : segs[0].lda;
        p.segk[i] = i < in.nseg ? in.seg_k[i] / 16 : 0;
        if (i < in.nseg && (segs[i].lda % 4 != 0 || ((uintptr_t)segs[i].ptr & 15) || segs[i].lda < in.seg_k[i]))
            throw Error(FC_ERR_INVALID, "launch_mlp_rows: A segment must be 16-byte aligned and at least as wide as its k range");
    }
    p.rowscal = rowscal;
    p.nlayers = 1 + (int)mid.size();
    p.ovf = flag;
    p.out16 = out16;
    p.out_s1 = out16_scale > 0.f ? out16_scale : 1.0f;
    p.out_s2 = out16_scale > 0.f ? 1.0f : 2048.0f;
    p.range_limit = 65504.0f / p.out_s1;
    p.rows_valid = rows_valid > 0 ? rows_valid : rows_alloc;
    for (int i = 0; i < 3; ++i) p.hbuf[i] = reinterpret_cast<unsigned short*>(h[i]);
    const int ks0 = mlp_rows_ks_pad(in.K_pad);
    p.L[0] = MlpRowsLayer{in.Wf, in.bias, in.colvec, ks0, -1, -1, 0};
    int cur = 0, keep = -1;                        // the rotation of run_mlp_hidden_generic (hostpack.cpp)
    double macs = (double)(in.k_true ? in.k_true : in.K_pad) * (in.n_true ? in.n_true : in.N_pad);
    for (size_t i = 0; i < mid.size(); ++i) {
        if (i % 2 == 0) keep = cur;
        int nxt = 0;
        while (nxt == cur || nxt == keep) ++nxt;
        p.L[i + 1] = MlpRowsLayer{mid[i].Wf, mid[i].bias, nullptr, MR_HID / 16, (i % 2 == 1) ? keep : -1, cur, nxt};
        macs += (double)(mid[i].k_true ? mid[i].k_true : mid[i].K_pad) * (mid[i].n_true ? mid[i].n_true : mid[i].N_pad);
        cur = nxt;
    }
    if (g_gemm_stamp == 4) {
        p.stamps = gemm_stamp_buffer(2 * 256 * 8);
        FC_HIP(hipMemsetAsync(p.stamps, 0, 2 * 256 * 8 * sizeof(unsigned long long), s));
    }
    const double flops = 2.0 * (double)(rows_valid > 0 ? rows_valid : rows_alloc) * macs;
    if (ks0 == 16 && p.stamps) mr_launch<16, FC_ACT_GELU, true>(p, rows_alloc, flops, s);        // (the stamped build exists for the C2 / C4 shape only)
    else if (ks0 == 16) mr_launch<16, FC_ACT_GELU>(p, rows_alloc, flops, s);
    else if (ks0 == 24) mr_launch<24, FC_ACT_GELU>(p, rows_alloc, flops, s);
    else mr_launch<32, FC_ACT_GELU>(p, rows_alloc, flops, s);
}

}  // namespace fc
