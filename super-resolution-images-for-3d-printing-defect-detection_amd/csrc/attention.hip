// attention.hip -- streaming-softmax core of the reference's SelfAttention layer
// (ESRGAN_model.py:57-65): s = g.f^T over N = H*W tokens, beta = softmax(s, axis=-1), o = beta.h,
// with d_qk = C/8 = 8 and d_v = C/2 = 32, no 1/sqrt(d) scale.  The N x N score matrix never
// reaches HBM (the reference materialises it, which is why it cannot leave 48x48 patches).
//
// Layout.  Transposed formulation so the softmax axis (keys) lies along a lane's registers:
//   S^T[key][query] = K.Q^T   (MFMA A = key fragment, B = query fragment)
//   O^T[dv][query] += V^T[dv][key] . P^T[key][query]   (A = V^T fragment from LDS, B = P^T straight
//   from the S^T accumulator registers -- no lane movement, see the accumulator-as-operand rule).
// A lane owns one query (column) and 16 of a 32-key tile's keys; the other 16 sit in lane+32,
// so the running max needs a single cross-half exchange per tile.
//
// Work split: workgroup = 4 waves = 128 queries of one image; every wave walks all keys in tiles
// of 64; the V tile is transposed into LDS once per workgroup per tile.
// HBM bytes per launch (algorithmic): B*N*(8+8+32+32)*sizeof(T); FLOP 2*B*N^2*(8+32);
// exp count B*N^2 (the binding resource: d_qk/d_v are tiny).
#include "common.h"

#include <cstdlib>

namespace {

constexpr int KV_TILE = 64;

template <typename T> struct AT;
template <> struct AT<float>  { static constexpr int PITCH = KV_TILE * 4 + 4; };    // bytes per V^T row in LDS

struct AttnParams {
    const char* qkv; int64_t cs; int qoff, koff, voff;
    char* o; int64_t o_cs; int o_coff;
    int N;
    const float* knorm;       // bf16 kernel: [B][ceil(N / 128)] largest Euclidean norm of a key in each 128-key group (attn_knorm_kernel)
};

__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32); }

// fp32 kernel (reference precision): exact fp32 MFMA (32x32x2_f32) for both products, one 32-query block per wave.
__global__ void __launch_bounds__(256, 2) attn_f32_kernel(AttnParams p) {
    constexpr int PITCH = AT<float>::PITCH;
    __shared__ __attribute__((aligned(16))) char vt[32 * PITCH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int N = p.N;
    const float* base = reinterpret_cast<const float*>(p.qkv) + (int64_t)b * N * p.cs;
    const int q = blockIdx.x * 128 + wave * 32 + r;
    const int qc = q < N ? q : N - 1;

    // query fragment (B operand of S^T = K.Q^T)
    float qf[4];
    {
        const float* qp = base + (int64_t)qc * p.cs + p.qoff;
#pragma unroll
        for (int j = 0; j < 4; ++j) qf[j] = qp[2 * j + h];
    }

    f32x16 oacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // V staging assignment: thread -> (key within tile, 8-channel octet)
    const int skey = tid >> 2, soct = tid & 3;

    for (int k0 = 0; k0 < N; k0 += KV_TILE) {
        __syncthreads();   // previous tile's V^T reads are done
        {
            const int key = k0 + skey;
            const float* vp = base + (int64_t)(key < N ? key : N - 1) * p.cs + p.voff + soct * 8;
            f32x4 v0 = *reinterpret_cast<const f32x4*>(vp), v1 = *reinterpret_cast<const f32x4*>(vp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                *reinterpret_cast<float*>(vt + (soct * 8 + e) * PITCH + skey * 4) = v0[e];
                *reinterpret_cast<float*>(vt + (soct * 8 + 4 + e) * PITCH + skey * 4) = v1[e];
            }
        }
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < KV_TILE / 32; ++sub) {
            const int kb = k0 + sub * 32;
            if (kb >= N) break;
            // ---- S^T = K . Q^T
            f32x16 s;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = 0.f;
            {
                const int key = kb + r;
                const float* kp = base + (int64_t)(key < N ? key : N - 1) * p.cs + p.koff;
#pragma unroll
                for (int j = 0; j < 4; ++j) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * j + h], qf[j], s, 0, 0, 0);
            }
            // ---- online softmax over this tile's 32 keys (16 here, 16 in lane^32)
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = kb + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (key >= N) s[i] = -INFINITY;
                mx = fmaxf(mx, s[i]);
            }
            mx = fmaxf(mx, xhalf(mx));
            const float m_new = fmaxf(m_run, mx);          // finite: every tile has >= 1 live key
            const float alpha = __expf(m_run - m_new);     // exp(-inf) = 0 on the first tile
            float psum = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] = __expf(s[i] - m_new); psum += s[i]; }
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[e] *= alpha;
            // ---- O^T += V^T . P^T
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int krow = sub * 32 + (i & 3) + 8 * (i >> 2);
                const float va = *reinterpret_cast<const float*>(vt + r * PITCH + (krow + 4 * h) * 4);
                oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(va, s[i], oacc, 0, 0, 0);
            }
        }
    }
    const float l_tot = l_run + xhalf(l_run);
    const float inv = 1.f / l_tot;
    if (q < N) {
        float* op = reinterpret_cast<float*>(p.o) + ((int64_t)b * N + q) * p.o_cs + p.o_coff;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = 8 * g + 4 * h;
            f32x4 t = {oacc[4 * g] * inv, oacc[4 * g + 1] * inv, oacc[4 * g + 2] * inv, oacc[4 * g + 3] * inv};
            *reinterpret_cast<f32x4*>(op + d0) = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 production kernel.  Same formulation; tuned for the binding resource (VALU / v_exp_f32 issue):
//   * QB 32-query blocks per wave (template; QB = 1 is the default: 86 VGPRs -> 5 waves/SIMD measured 10 % faster than
//     QB = 2, which shares every key/V^T fragment between two blocks but drops to 2 waves/SIMD); key tile = 128 keys
//     per barrier;
//   * softmax in the exp2 domain: p = exp2(s*log2e - m'), one fma + one v_exp_f32 per score;
//   * the O/l rescale runs only when some lane's running max actually grew (wave-uniform vote), which
//     after the first few key tiles is rare; the result is bit-identical to rescaling every tile
//     because exp2(0) = 1 exactly;
//   * key masking (-inf) only on the ragged last tile; key fragments are fetched one tile ahead;
//   * V is transposed into LDS two keys per dword (8 ds_write_b32 per thread per 128 keys).
// ------------------------------------------------------------------------------------------------
constexpr int KT2 = 128;                       // keys per barrier
constexpr int PITCH2 = KT2 * 2 + 8;            // bytes per V^T row: 66 dwords -> conflict-free ds_read_b64

// Largest |k| of every 128-key group, one wave per group.  It lets the main kernel decide per GROUP, with scalar arithmetic only, that no
// score of the group can outgrow the running max by more than the rescale-free bound (Cauchy-Schwarz: k.q <= |k| |q|), and then run tiles
// that carry no max chain and no vote at all.
__global__ void __launch_bounds__(64) attn_knorm_kernel(const bf16_t* qkv, int64_t cs, int koff, int N, int ngroups, float* out) {
    const int g = blockIdx.x % ngroups, b = blockIdx.x / ngroups, lane = threadIdx.x;
    const bf16_t* base = qkv + (int64_t)b * N * cs + koff;
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int key = g * 128 + t * 64 + lane;
        if (key < N) {
            const bf16x8 k = *reinterpret_cast<const bf16x8*>(base + (int64_t)key * cs);
            float n2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) n2 += (float)k[e] * (float)k[e];
            m = fmaxf(m, n2);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) out[(int64_t)b * ngroups + g] = sqrtf(m);
}

// __launch_bounds__(256, 3) (second argument: waves per SIMD): with a register ceiling <= 256 hipcc keeps MFMA results in VGPRs.
// Without it the score tile and the output accumulator shared one AGPR block and every key tile paid 16 v_accvgpr_read + 32
// v_accvgpr_write (ISA count), a third of the VALU slots of this VALU-bound loop.  3 waves/SIMD = 168 registers: the two query
// blocks need 171 unconstrained, the cap costs one 8-byte spill outside the key loop and is 12 % faster than 2 waves/SIMD.
template <int QB>
__global__ void __launch_bounds__(256, 3) attn_bf16_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) char vt[32 * PITCH2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int N = p.N;
    constexpr int QW = 64 * QB * 2;   // queries per workgroup (4 waves x QB blocks of 32)
    const int nqt = (N + QW - 1) / QW;
    const int b = blockIdx.x / nqt, qt = blockIdx.x - b * nqt;
    const bf16_t* base = reinterpret_cast<const bf16_t*>(p.qkv) + (int64_t)b * N * p.cs;
    const int cs = (int)p.cs;
    int qi[QB];
    bf16x8 qb[QB];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        qi[j] = qt * QW + wave * (32 * QB) + j * 32 + r;
        const int qc = qi[j] < N ? qi[j] : N - 1;
        bf16x8 z = {};
        qb[j] = z;                                     // h == 1 lanes: K-slots 8..15; slots 8, 9 will carry the running max
        if (h == 0) qb[j] = *reinterpret_cast<const bf16x8*>(base + (int64_t)qc * cs + p.qoff);
    }
    // |q| of the wave's largest query (wave-uniform): with attn_knorm's |k| it bounds every score of a key group
    float qn2 = 0.f;
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        float n2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) n2 += (float)qb[j][e] * (float)qb[j][e];
        qn2 = fmaxf(qn2, n2);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) qn2 = fmaxf(qn2, __shfl_xor(qn2, o));
    const float qmaxv = sqrtf(qn2) * 1.0009765625f;      // wave-uniform (a 2^-10 margin over the fp32 rounding of norms and products)
    const int* knorm_bits = reinterpret_cast<const int*>(p.knorm) + (int64_t)b * ((N + KT2 - 1) / KT2);
    f32x16 oacc[QB];
    // Row sums ride on the matrix core as well: a 16x16x32 MFMA with a 0/1 selector as A and the probability fragment (the PV
    // MFMA's B operand, reinterpreted) as B.  Lane (r, h) of the 32x32 layout is column r & 15, k-group 2h + (r >> 4) of the
    // 16x16x32 layout, so selector row 0 = ones on k-groups {0, 2} sums queries 0..15 and row 1 = ones on {1, 3} sums queries
    // 16..31: lacc[j][0] / [1] in lanes 0..15 hold the running denominators of queries n / n + 16.  (The sums are then taken
    // over the same bf16-rounded probabilities that enter the numerator.)
    bf16x8 sel = {};
    {
        const int m = lane & 15, kg = lane >> 4;
        const bf16_t one = (bf16_t)1.f;
        if ((m == 0 && (kg & 1) == 0) || (m == 1 && (kg & 1) == 1)) sel = bf16x8{one, one, one, one, one, one, one, one};
    }
    f32x4 lacc[QB];
    float m_run[QB];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        m_run[j] = 0.f; lacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};   // the first key tile sets m_run unconditionally (first = true)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[j][e] = 0.f;
    }

    const int kp = tid >> 2, oct = tid & 3;    // V staging: key pair, 8-channel octet
    // Key fragment (A operand): lanes h == 0 hold the 8 key dims, lanes h == 1 hold K-slots 8..15 = {-1, -1, -1, 0, ...}.  With the
    // query fragment's slots 8..10 = the running max split into three bf16 pieces (exact to fp32), the score MFMA returns  k.q - m  directly: the running max is subtracted
    // inside the matrix core and the softmax loop has no multiply/subtract per score left.  (k is pre-scaled by log2(e) in the
    // projection conv, api.hip, so the scores arrive in the exp2 domain.)
    bf16x8 kneg = {};
    kneg[0] = (bf16_t)-1.f; kneg[1] = (bf16_t)-1.f; kneg[2] = (bf16_t)-1.f;
    auto load_k = [&](int kb) {
        // the lane number is re-derived here (two v_mbcnt) rather than kept: at the 168-register budget hipcc spilled the copy of r this address
        // needs, and the reload's s_waitcnt vmcnt(0) at the head of every key group also waited for the key fragment requested just before
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        bf16x8 kf = kneg;
        const int key = kb + (l & 31);
        if (l < 32) kf = *reinterpret_cast<const bf16x8*>(base + (int64_t)(key < N ? key : N - 1) * cs + p.koff);
        return kf;
    };
    bf16x8 knext = load_k(0);
    f32x16 zero16;                                     // loop-invariant C operand of the score MFMA (never written)
#pragma unroll
    for (int e = 0; e < 16; ++e) zero16[e] = 0.f;
    struct { } false_c; struct { int x; } true_c{0};
    // max over the two lane halves without the LDS crossbar: v_permlane32_swap is a plain VALU op
    auto hmax = [](float v) {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        return fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
    };
    auto tile = [&](const bf16x8 kf, const int sub, const int kb, auto ragged_tag, const bool first) {
        constexpr bool RAGGED = sizeof(ragged_tag) > 1;
        // phase A (both query blocks, no control flow in between): scores relative to the running max, and their column maxima
        f32x16 s[QB];
        float mx[QB];
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            s[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qb[j], zero16, 0, 0, 0);
            if constexpr (RAGGED) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (kb + (i & 3) + 8 * (i >> 2) + 4 * h >= N) s[j][i] = -INFINITY;
            }
            float m = fmaxf(fmaxf(s[j][0], s[j][1]), s[j][2]);
#pragma unroll
            for (int i = 3; i + 1 < 16; i += 2) m = fmaxf(fmaxf(m, s[j][i]), s[j][i + 1]);
            m = fmaxf(m, s[j][15]);
            mx[j] = hmax(m);                            // > 0: a score above the running max
        }
        // phase B: one wave-uniform vote for both blocks; only when a running max grew (or on the first tile) are the
        // accumulators rescaled, the scores shifted and the max pieces in the query fragment refreshed
        bool grew = first;
#pragma unroll
        for (int j = 0; j < QB; ++j) grew = grew || mx[j] > 0.f;
        if (__any(grew)) {
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                const float delta = first ? mx[j] : fmaxf(mx[j], 0.f);
                const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);   // first tile: l_run = oacc = 0, and exp2(-delta) may be +inf
                lacc[j][0] *= alpha;                                   // lanes 0..15: query n is this lane's own query ...
                lacc[j][1] *= __shfl(alpha, (lane + 16) & 63);        // ... and query n + 16 is lane n + 16's
#pragma unroll
                for (int e = 0; e < 16; ++e) oacc[j][e] *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) s[j][i] -= delta;
                m_run[j] += delta;
                if (h == 1) {                                             // m = m1 + m2 + m3: three bf16 pieces carry all 24 bits
                    const bf16_t m1 = (bf16_t)m_run[j];
                    const float r1 = m_run[j] - (float)m1;
                    const bf16_t m2 = (bf16_t)r1;
                    qb[j][0] = m1; qb[j][1] = m2; qb[j][2] = (bf16_t)(r1 - (float)m2);
                }
            }
        }
        // phase C (both blocks interleavable): probabilities; O^T += V^T . P^T and the row sums on the matrix core
#pragma unroll
        for (int j = 0; j < QB; ++j) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[j][i] = __builtin_amdgcn_exp2f(s[j][i]);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const char* vrow = vt + r * PITCH2 + (sub * 32 + 16 * ks + 4 * h) * 2;
            const bf16x4 va = *reinterpret_cast<const bf16x4*>(vrow);
            const bf16x4 vb = *reinterpret_cast<const bf16x4*>(vrow + 16);
            const bf16x8 vf = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                bf16x8 pf;
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[e] = (bf16_t)s[j][8 * ks + e];
                oacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[j], 0, 0, 0);
                lacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sel, pf, lacc[j], 0, 0, 0);
            }
        }
    };

    // Fast tile: a full tile whose scores stay within 2^GROW_OK of the running max needs no rescale at all -- the probabilities are
    // then simply taken relative to the (stale) running max, which is exact: numerator and denominator carry the same power of two.
    // It returns false, before touching any state, when some score exceeds that bound; the caller then redoes the tile with the
    // full logic above.  The point is what the fast loop does NOT contain: with the rescale branch inside the key loop, hipcc
    // joins the branch's new accumulators / denominators / query fragments with the untouched ones in a phi and copies all 48
    // registers on the common path of every tile (24 v_mov_b64), and the loop also pays the vote; a loop made of fast tiles only
    // has neither (attention 241 -> ~185 ms per two bench steps).
    constexpr float GROW_OK = 40.f;    // 2^40 x 9216 keys x |v|: far inside fp32 for the sums and the bf16 probabilities
    // guard (wave-uniform) = false: the group-level bound below already says that no score of this group exceeds the running max of any of
    // the wave's queries by more than 2^GROW_OK -- the tile then carries no max chain and no vote: score MFMA, 16 exps, 8 converts, 3 MFMAs.
    auto tile_fast = [&](const bf16x8 kf, const int sub, const bool guard) -> bool {
        f32x16 s[QB];
#pragma unroll
        for (int j = 0; j < QB; ++j) s[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qb[j], zero16, 0, 0, 0);
        if (guard) {
            bool risky = false;
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                float m = fmaxf(fmaxf(s[j][0], s[j][1]), s[j][2]);
#pragma unroll
                for (int i = 3; i + 1 < 16; i += 2) m = fmaxf(fmaxf(m, s[j][i]), s[j][i + 1]);
                m = fmaxf(m, s[j][15]);
                risky = risky || m > GROW_OK;
            }
            if (__any(risky)) return false;
        }
#pragma unroll
        for (int j = 0; j < QB; ++j) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[j][i] = __builtin_amdgcn_exp2f(s[j][i]);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const char* vrow = vt + r * PITCH2 + (sub * 32 + 16 * ks + 4 * h) * 2;
            const bf16x4 va = *reinterpret_cast<const bf16x4*>(vrow);
            const bf16x4 vb = *reinterpret_cast<const bf16x4*>(vrow + 16);
            const bf16x8 vf = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                bf16x8 pf;
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[e] = (bf16_t)s[j][8 * ks + e];
                oacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[j], 0, 0, 0);
                lacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sel, pf, lacc[j], 0, 0, 0);
            }
        }
        return true;
    };
    // The group-level bound as ONE scalar integer compare per group.  |k|max |q|max - min m <= GROW_OK  <=>  |k|max <= (GROW_OK + min m) / |q|max =:
    // thresh, and for non-negative floats the order of the values is the order of their bit patterns: thresh is kept as its bits in a scalar
    // register (negative -- bound unreachable -- as -1.f, whose bits are a negative integer; +inf when every query is zero), refreshed only
    // when the smallest running max among the wave's queries may have changed (after a group that ran the full logic).  gfx950 has no scalar
    // float ALU: the first version evaluated the bound in vector registers per group, hipcc spilled |q|max at the 168-register budget, and the
    // reload's s_waitcnt vmcnt(0) also waited for the key fragment just requested.
    auto thresh_of_wave = [&]() -> int {
        float m = m_run[0];
#pragma unroll
        for (int j = 1; j < QB; ++j) m = fminf(m, m_run[j]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o));
        const float lim = GROW_OK + m;
        const float th = !(lim >= 0.f) ? -1.f : (qmaxv > 0.f ? lim / qmaxv * 0.9990234375f : INFINITY);
        return __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, th));      // an integer builtin: the float travels as its bits
    };
    int thresh_bits = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, -1.f));      // until the first group has set the running maxima
    // V^T of KT2 keys into LDS, two keys per dword (every wave of the workgroup passes here once per key group, in either mode).
    // (Round 3 measured a double-buffered image with the rows requested a whole group ahead: no faster -- the loop is bound by vector
    // issue, 16 v_exp_f32 + 8 converts + 5 MFMAs per 32 x 32 scores, not by this staging -- and eight more live registers.)
    auto stage_v = [&](const int k0) {
        __syncthreads();
#pragma unroll
        for (int part = 0; part < KT2 / 128; ++part) {
            const int ka = k0 + part * 128 + 2 * kp, kbb = ka + 1;
            if (ka - 2 * kp >= N) break;                  // this 128-key part lies wholly beyond the keys (workgroup-uniform)
            const bf16x8 va = *reinterpret_cast<const bf16x8*>(base + (int64_t)(ka < N ? ka : N - 1) * cs + p.voff + oct * 8);
            const bf16x8 vb = *reinterpret_cast<const bf16x8*>(base + (int64_t)(kbb < N ? kbb : N - 1) * cs + p.voff + oct * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
                bf16x2 pr = {va[e], vb[e]};
                *reinterpret_cast<bf16x2*>(vt + (oct * 8 + e) * PITCH2 + (part * 64 + kp) * 4) = pr;
            }
        }
        __syncthreads();
    };
    // the tiles [sub0, KT2/32) of key group k0 with the full logic
    auto slow_tiles = [&](const int k0, const int sub0, bf16x8 kf) {
        for (int sub = sub0; sub < KT2 / 32; ++sub) {
            const int kb = k0 + sub * 32;
            if (kb >= N) break;
            if (sub != sub0) { kf = knext; knext = load_k(kb + 32); }
            if (kb + 32 <= N) tile(kf, sub, kb, false_c, kb == 0);  // full tile: no masking code at all
            else tile(kf, sub, kb, true_c, kb == 0);                // ragged last tile: keys >= N get -inf
        }
    };

    // Key groups: fast tiles while they last, the full logic for the rest of the group (the first group, which sets the running max,
    // the ragged tail, and any group in which a score outgrew the bound).  The conditional slow part sits once per GROUP, so the
    // copies its phi costs are paid once per four tiles instead of in every tile.
    for (int k0 = 0; k0 < N; k0 += KT2) {
        stage_v(k0);
        int sub = 0;
        bf16x8 kf = knext;
        const bool full = k0 != 0 && k0 + KT2 <= N;
        if (full) {
            // group-level bound (scalar): every score k.q - m of this group is <= |k|max |q|max - min m; where it holds the tiles run unguarded
            const bool guard = !(knorm_bits[k0 / KT2] <= thresh_bits);       // NaN norms have bits above every finite threshold: guarded
#pragma unroll
            for (; sub < KT2 / 32; ++sub) {
                kf = knext;
                knext = load_k(k0 + sub * 32 + 32);          // next tile's keys fly under this tile's softmax
                if (!tile_fast(kf, sub, guard)) break;
            }
        } else {
            knext = load_k(k0 + 32);
        }
        if (sub < KT2 / 32) {
            slow_tiles(k0, sub, kf);
            thresh_bits = thresh_of_wave();
        }
    }
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        const float l0 = __shfl(lacc[j][0], r & 15), l1 = __shfl(lacc[j][1], r & 15);
        const float inv = 1.f / ((r >> 4) ? l1 : l0);
        if (qi[j] < N) {
            bf16_t* op = reinterpret_cast<bf16_t*>(p.o) + ((int64_t)b * N + qi[j]) * p.o_cs + p.o_coff;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 t = {(bf16_t)(oacc[j][4 * g] * inv), (bf16_t)(oacc[j][4 * g + 1] * inv), (bf16_t)(oacc[j][4 * g + 2] * inv),
                            (bf16_t)(oacc[j][4 * g + 3] * inv)};
                *reinterpret_cast<bf16x4*>(op + 8 * g + 4 * h) = t;
            }
        }
    }
}

}  // namespace

int attention_launch(sr_ctx* ctx, int dtype, const void* qkv, int64_t cs, int qoff, int koff, int voff, int B, int N,
                     void* o, int64_t o_cs, int o_coff, hipStream_t st) {
    if (B <= 0 || N <= 0) return ctx->fail(SR_ERR_INVALID, "attention: empty tensor");
    const int esz = dtype_size(dtype);
    if ((cs * esz) % 16 || (qoff * esz) % 16 || (koff * esz) % 16 || (voff * esz) % 16 || (o_cs * esz) % 16 || (o_coff * esz) % 16)
        return ctx->fail(SR_ERR_INVALID, "attention: views must be 16-byte aligned");
    AttnParams p{static_cast<const char*>(qkv), cs, qoff, koff, voff, static_cast<char*>(o), o_cs, o_coff, N, nullptr};
    dim3 grid((unsigned)((N + 127) / 128), (unsigned)B);
    if (dtype != SR_DTYPE_BF16 && dtype != SR_DTYPE_F32) return ctx->fail(SR_ERR_INVALID, "attention: dtype must be f32 or bf16");
    const int rec = ctx->prof_open(dtype == SR_DTYPE_BF16 ? "attn<bf16>" : "attn<f32>", 2.0 * B * (double)N * N * 40.0,
                                   (double)B * N * 80.0 * esz, st);
    if (dtype == SR_DTYPE_BF16) {
        // two 32-query blocks per wave (256 queries per workgroup): the blocks share the key fragment and give the
        // scheduler two independent softmax chains (measured 3 % faster than one block per wave)
        static const int qb_env = [] { const char* e = getenv("SR355_ATTN_QB"); return e ? atoi(e) : 2; }();      // tuning switch: query blocks per wave
        const int QBr = qb_env == 1 ? 1 : 2;
        const int64_t nwg = (int64_t)B * ((N + 128 * QBr - 1) / (128 * QBr));
        const int ngroups = (N + KT2 - 1) / KT2;
        if (nwg >= (1ll << 31) || (int64_t)B * ngroups >= (1ll << 31)) { ctx->prof_close(rec, st); return ctx->fail(SR_ERR_INVALID, "attention: too many workgroups for one launch"); }
        float* kn = static_cast<float*>(ctx->arena(ctx->attn_kn, sizeof(float) * (size_t)B * ngroups, st));
        if (!kn) { ctx->prof_close(rec, st); return SR_ERR_OOM; }
        p.knorm = kn;
        hipLaunchKernelGGL(attn_knorm_kernel, dim3((unsigned)((int64_t)B * ngroups)), dim3(64), 0, st, reinterpret_cast<const bf16_t*>(qkv) + 0, cs, koff, N, ngroups, kn);
        if (QBr == 1) hipLaunchKernelGGL(attn_bf16_kernel<1>, dim3((unsigned)nwg), dim3(256), 0, st, p);
        else hipLaunchKernelGGL(attn_bf16_kernel<2>, dim3((unsigned)nwg), dim3(256), 0, st, p);
    }
    else hipLaunchKernelGGL(attn_f32_kernel, grid, dim3(256), 0, st, p);
    ctx->prof_close(rec, st);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}
