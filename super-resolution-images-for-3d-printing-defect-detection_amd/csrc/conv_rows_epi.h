// conv_rows_epi.h -- epilogue of the row-sliding 3x3 kernels.  Lane (px, q) owns pixel column px and couts 4q..4q+3 of
// each 16-cout block; acc[r][n] is output row (y0 + wave*R + r), cout block n.
//
// The fast path is specialised at compile time on which skips exist and on the store shape, and is written so that the
// wave never waits for one of its own stores: every skip value is loaded first, then the rows are finished and stored
// back to back.  (With the skips behind run-time flags the compiler could not tell whether a load was outstanding and put
// s_waitcnt vmcnt(0) in front of every row -- each row then waited for the previous row's store to reach L2, ~1.5k
// cycles a row, which made the epilogue the longest phase of a workgroup: in-kernel stamps)
// Output addresses are a wave-uniform row base + a per-lane 32-bit offset computed once; rows advance the base only.
#pragma once
#include "conv_common.h"

namespace convk {

template <int NB16, int R, bool HAS1, bool HAS2, bool PAIR>
__device__ __forceinline__ void rows_epilogue_fast(const ConvParams& p, f32x4 (&acc)[R][NB16], const f32x4 (&biasv)[NB16], int b, int y0,
                                                   int x0, int ct, int wave, int px, int q) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int H = p.H, W = p.W;
    const int ox = x0 + px;
    const int oyw = y0 + wave * R;                         // first output row of this wave (wave-uniform)
    const int rows = min(R, H - oyw);                     // wave-uniform count of live rows
    if (rows <= 0) return;
    const int cell_h = p.cell_h, cell_w = p.cell_w;       // CellGrid separators (common.h): never stored, so they stay zero
    const bool col_ok = ox < W && (cell_w == 0 || ox % cell_w != cell_w - 1);
    const int oxc = ox < W ? ox : 0;
    // activation is branch-free: max(v, slope*v) with slope 1 (linear), 0 (relu), 0.2 (leaky relu)
    const float slope = p.act == SR_ACT_RELU ? 0.f : (p.act == SR_ACT_LRELU ? 0.2f : 1.f);
    const float alpha = p.alpha, beta1 = p.beta1, beta2 = p.beta2;
    const bool clip = p.clip != 0, of32 = p.out_f32 != 0;

    // ---- skip values are loaded a row group at a time, ahead of that group's stores: the whole tile at once, or two
    //      halves where both skips are present and the register budget is the 3-workgroups-per-CU one (NB16 <= 2)
    constexpr int RG = (HAS1 && HAS2 && NB16 <= 2 && R >= 2) ? R / 2 : R;
    int cc[NB16];
#pragma unroll
    for (int n = 0; n < NB16; ++n) cc[n] = min((ct * NB16 + n) * 16 + 4 * q, p.Cout - 4);

    // ---- store addressing: NS stores per row; store s of row r goes to element rb[s] + r*rstep (uniform) + loff[s] (lane)
    constexpr int NS = PAIR ? NB16 / 2 : NB16;
    const int rr_ = p.r, Cd = p.Cd;
    int loff[NS];
    int64_t rb[NS];
    bool live[NS];
    const int64_t rstep = rr_ <= 1 ? (int64_t)p.out_rs : (int64_t)rr_ * p.out_rs;   // (with d2s the host sets out_rs for the r-times wider output row)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        // PAIR: lanes q and q^1 trade halves (v_permlane16_swap): an even-q lane stores 8 consecutive couts of block 2s,
        // an odd-q lane 8 of block 2s+1, starting at cout 4*(q & ~1) of the block
        const int blk = (ct * NB16 + (PAIR ? 2 * s : s)) * 16;                 // wave-uniform
        const int lane_c = PAIR ? (q & 1) * 16 + 4 * (q & ~1) : 4 * q;
        live[s] = col_ok && (PAIR || blk + lane_c < p.Cout);
        if (rr_ <= 1) {
            rb[s] = ((int64_t)b * H + oyw) * p.out_rs;
            loff[s] = ox * (int)p.out_cs + choff(p.out_coff + blk + lane_c, p.out_ps);
        } else {   // TF depth_to_space "DCR": cout = (i*r + j)*Cd + c.  Cd % 16 == 0 (% 32 for PAIR): (i, j) uniform per store
            const int sub = blk / Cd, cb = blk - sub * Cd;
            const int i = sub / rr_, j = sub - i * rr_;
            rb[s] = (((int64_t)b * H + oyw) * rr_ + i) * p.out_rs;
            loff[s] = (ox * rr_ + j) * (int)p.out_cs + p.out_coff + cb + lane_c;
        }
    }

#pragma unroll
    for (int r0 = 0; r0 < R; r0 += RG) {
    if (r0 >= rows) break;                                  // wave-uniform
    bf16x4 k1[HAS1 ? RG : 1][NB16], k2[HAS2 ? RG : 1][NB16];
    if (HAS1 || HAS2) {
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            const int64_t row = (int64_t)b * H + (oyw + (r0 + g < rows ? r0 + g : 0));           // dead rows re-read row 0: valid memory
#pragma unroll
            for (int n = 0; n < NB16; ++n) {
                if (HAS1) k1[g][n] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(p.s1) + row * p.s1_rs + (oxc * (int)p.s1_cs + choff(p.s1_coff + cc[n], p.s1_ps)));
                if (HAS2) k2[g][n] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(p.s2) + row * p.s2_rs + (oxc * (int)p.s2_cs + choff(p.s2_coff + cc[n], p.s2_ps)));
            }
        }
    }
#pragma unroll
    for (int g = 0; g < RG; ++g) {
        const int r = r0 + g;
        if (r >= rows) break;                               // wave-uniform
        const bool row_ok = cell_h == 0 || (oyw + r) % cell_h != cell_h - 1;   // wave-uniform: a separator row is computed like any other, not stored
        f32x4 v[NB16];
#pragma unroll
        for (int n = 0; n < NB16; ++n) {
            v[n] = acc[r][n] + biasv[n];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[n][e] = fmaxf(v[n][e], v[n][e] * slope) * alpha;
            if (HAS1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[n][e] += beta1 * (float)k1[g][n][e];
            }
            if (HAS2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[n][e] += beta2 * (float)k2[g][n][e];
            }
            if (clip) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[n][e] = fminf(fmaxf(v[n][e], 0.f), 1.f);
            }
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int64_t base = rb[s] + r * rstep;
            if (PAIR) {
                const int n = 2 * s;
                const bf16x4 a = {(bf16_t)v[n][0], (bf16_t)v[n][1], (bf16_t)v[n][2], (bf16_t)v[n][3]};
                const bf16x4 c = {(bf16_t)v[n + 1][0], (bf16_t)v[n + 1][1], (bf16_t)v[n + 1][2], (bf16_t)v[n + 1][3]};
                const u32x2 au = __builtin_bit_cast(u32x2, a), cu = __builtin_bit_cast(u32x2, c);
                const auto s0 = __builtin_amdgcn_permlane16_swap(au[0], cu[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(au[1], cu[1], false, false);
                const u32x4 o = {(unsigned)s0[0], (unsigned)s1[0], (unsigned)s0[1], (unsigned)s1[1]};
                if (live[s] && row_ok) *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.out) + base + loff[s]) = o;
            } else if (of32) {
                if (live[s] && row_ok) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + base + loff[s]) = v[s];
            } else {
                const bf16x4 o = {(bf16_t)v[s][0], (bf16_t)v[s][1], (bf16_t)v[s][2], (bf16_t)v[s][3]};
                if (live[s] && row_ok) *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.out) + base + loff[s]) = o;
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // finish a row before starting the next: keeps the register peak at one row of temporaries
    }
    }
}

// Which epilogue a launch takes (wave-uniform, fixed for the whole kernel): -1 = the generic per-element path (odd channel counts /
// unaligned views / tanh: the final RGB conv), else 2*skips + pair with skips = 0 (none), 1 (skip 1), 2 (both) and pair = 16-byte
// paired stores.  The host passes a lone skip as s1 (conv_launch swaps), so "s2 only" does not occur.
template <int NB16>
__device__ __forceinline__ int rows_epilogue_kind(const ConvParams& p) {
    const int Cd = p.Cd, rr_ = p.r;
    const bool has1 = p.s1 != nullptr, has2 = p.s2 != nullptr;
    const bool fast = p.vec != 0 && (p.Cout & 3) == 0 && p.act != SR_ACT_TANH && (rr_ <= 1 || (Cd & 15) == 0) && (has1 || !has2);
    if (!fast) return -1;
    const bool pair_ok = NB16 >= 2 && p.out_f32 == 0 && (rr_ <= 1 || (Cd & 31) == 0) && (p.Cout & 31) == 0;
    return 2 * (has2 ? 2 : (has1 ? 1 : 0)) + (pair_ok ? 1 : 0);
}

// One epilogue, chosen at compile time (a kernel that runs the epilogue inside a loop instantiates its loop per KIND, so that
// only that variant's loop invariants stay live).
template <int NB16, int R, int KIND>
__device__ __forceinline__ void rows_epilogue_as(const ConvParams& p, f32x4 (&acc)[R][NB16], const f32x4 (&biasv)[NB16], int b, int y0, int x0,
                                                 int ct, int wave, int px, int q) {
    if constexpr (KIND < 0) {
        const int H = p.H, W = p.W;
        const int ox = x0 + px;
        const int oyw = y0 + wave * R;
        if (ox < W) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int oy = oyw + r;
                if (oy >= H) continue;
#pragma unroll
                for (int n = 0; n < NB16; ++n) {
                    const float a[4] = {acc[r][n][0], acc[r][n][1], acc[r][n][2], acc[r][n][3]};
                    epilogue4<bf16_t>(p, b, oy, ox, (ct * NB16 + n) * 16 + 4 * q, a);
                }
            }
        }
    } else {
        constexpr bool PAIR = (KIND & 1) != 0 && NB16 >= 2;
        rows_epilogue_fast<NB16, R, (KIND >> 1) >= 1, (KIND >> 1) >= 2, PAIR>(p, acc, biasv, b, y0, x0, ct, wave, px, q);
    }
}

// Run-time dispatch over the kinds (one-shot use at the end of a tile kernel).
template <int NB16, int R>
__device__ __forceinline__ void rows_epilogue(const ConvParams& p, f32x4 (&acc)[R][NB16], const f32x4 (&biasv)[NB16], int b, int y0, int x0,
                                              int ct, int wave, int px, int q) {
    switch (rows_epilogue_kind<NB16>(p)) {
        case -1: rows_epilogue_as<NB16, R, -1>(p, acc, biasv, b, y0, x0, ct, wave, px, q); break;
        case 0: rows_epilogue_as<NB16, R, 0>(p, acc, biasv, b, y0, x0, ct, wave, px, q); break;
        case 1: rows_epilogue_as<NB16, R, 1>(p, acc, biasv, b, y0, x0, ct, wave, px, q); break;
        case 2: rows_epilogue_as<NB16, R, 2>(p, acc, biasv, b, y0, x0, ct, wave, px, q); break;
        case 3: rows_epilogue_as<NB16, R, 3>(p, acc, biasv, b, y0, x0, ct, wave, px, q); break;
        case 4: rows_epilogue_as<NB16, R, 4>(p, acc, biasv, b, y0, x0, ct, wave, px, q); break;
        default: rows_epilogue_as<NB16, R, 5>(p, acc, biasv, b, y0, x0, ct, wave, px, q); break;
    }
}

}  // namespace convk
