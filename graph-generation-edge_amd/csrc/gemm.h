// gemm.h - fp32 MFMA building blocks (v_mfma_f32_32x32x2_f32, exact fp32 fma chains).
//
// Operand scheme.  One wave owns a 32-row x 32-col output tile (16 accumulator registers).  The
// instruction sums over k in {0,1}: lanes 0-31 supply k=0, lanes 32-63 supply k=1.  Which PHYSICAL
// k a lane feeds is free as long as A and B agree, so for a 64-deep chunk lane half h feeds
// k = 32*h + s at step s (s = 0..31).  Each lane then needs 32 CONTIGUOUS floats of its A row and
// of its W row (torch Linear stores W as [N,K], k contiguous): eight 16-byte loads, no LDS, no
// transposes.  C layout: lane l holds column l&31, register r holds row (r&3) + 8*(r>>2) + 4*(l>>5).
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct eh_frag { f32x4 v[8]; };   // 32 floats of one row, one k-half of a 64-deep chunk

// p points at row_base + chunk_k0 + 32*half ; must be 16-byte aligned
__device__ static inline void eh_load_frag(eh_frag& f, const float* __restrict__ p) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f.v[i] = *reinterpret_cast<const f32x4*>(p + 4 * i);
}

__device__ static inline void eh_mma64(f32x16& acc, const eh_frag& a, const eh_frag& b) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[i][c], b.v[i][c], acc, 0, 0, 0);
    }
}

__device__ static inline int eh_crow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

static inline __device__ f32x16 eh_zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.0f;
    return z;
}

// ---- 64-wide fp32 tiles in LDS -----------------------------------------------------------------
// Row pitch 68 floats: a lane reads its 32-float fragment with eight ds_read_b128 and the 16 lanes of a
// b128 group hit 16 different 4-bank slots (conflict-free).
#define EH_WPITCH 68
__device__ static inline void eh_load_frag_lds(eh_frag& f, const float* tile, int row, int hh) {
    const float* p = tile + row * EH_WPITCH + hh * 32;
#pragma unroll
    for (int i = 0; i < 8; ++i) f.v[i] = *reinterpret_cast<const f32x4*>(p + 4 * i);
}

// ---- weights in fragment order (k_pack_w in gnn.hip) ---------------------------------------------
// tile (nt, kc) of a packed [N][K] matrix starts at ((nt * K/64) + kc) * 2048 floats; a fragment is
// eight fully coalesced 1-KiB loads.
__device__ static inline void eh_load_frag_packed(eh_frag& f, const float* __restrict__ packed, int KC, int nt, int kc,
                                                  int lane) {
    const float* p = packed + ((size_t)(nt * KC + kc) * 2048) + lane * 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) f.v[i] = *reinterpret_cast<const f32x4*>(p + i * 256);
}

// ---- fp32 products on the bf16 matrix pipe ------------------------------------------------------
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the rate of v_mfma_f32_32x32x16_bf16.  An fp32 value splits EXACTLY
// into three bf16 terms (x = h + m + l: 8 + 8 + 8 significant bits, by truncation), so the fp32 product sum
// is the sum of nine bf16 product sums; the three smallest (m*l, l*m, l*l <= 2^-24 relative, below fp32
// rounding) are dropped and six remain: 6 x 4 MFMAs of 32 cycles per 64-deep chunk instead of 32 of 64 cycles
// (2.7x fewer matrix-pipe cycles), fp32 accumulation as before, and the SAME accumulator layout (eh_crow), so
// epilogues are unchanged.  Measured error against fp64 is not larger than the fp32 instruction's.
// Weights are split once (k_pack_w), activations per fragment (eh_split3: 5.5 vector ops per value).
// Operand mapping: the bf16 instruction takes k = 8*(lane>>5) + j per lane; which physical k that is is free as
// long as A and B agree, so group i (0..3) of a fragment feeds its k = 32*(lane>>5) + 8i + j.
#ifndef EH_BF16X3
#define EH_BF16X3 1
#endif
typedef __bf16 eh_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int eh_u32x4 __attribute__((ext_vector_type(4)));
struct eh_frag3 { eh_u32x4 h[4], m[4], l[4]; };      // [i]: k-group 8i..8i+7 of the lane's 32 k, bf16 pairs in dwords

__device__ static inline void eh_split_pair(float x0, float x1, unsigned& ph, unsigned& pm, unsigned& pl) {
    typedef float eh_f2 __attribute__((ext_vector_type(2)));
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    // the two remainders of a pair as ONE packed subtraction each (exact either way)
    const eh_f2 r = eh_f2{x0, x1} - eh_f2{__uint_as_float(u0 & 0xffff0000u), __uint_as_float(u1 & 0xffff0000u)};
    const unsigned v0 = __float_as_uint(r[0]), v1 = __float_as_uint(r[1]);
    const eh_f2 t = r - eh_f2{__uint_as_float(v0 & 0xffff0000u), __uint_as_float(v1 & 0xffff0000u)};
    ph = __builtin_amdgcn_perm(u1, u0, 0x07060302u);          // upper halves: element 0 in the low 16 bits
    pm = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    pl = __builtin_amdgcn_perm(__float_as_uint(t[1]), __float_as_uint(t[0]), 0x07060302u);
}
__device__ static inline void eh_split3(eh_frag3& o, const eh_frag& f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            unsigned ph, pm, pl;
            eh_split_pair(f.v[2 * i + (p >> 1)][2 * (p & 1)], f.v[2 * i + (p >> 1)][2 * (p & 1) + 1], ph, pm, pl);
            o.h[i][p] = ph; o.m[i][p] = pm; o.l[i][p] = pl;
        }
    }
}
#define EH_MFMA_BF16(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(eh_bf16x8, A), __builtin_bit_cast(eh_bf16x8, B), C, 0, 0, 0)
__device__ static inline void eh_mma64(f32x16& acc, const eh_frag3& a, const eh_frag3& b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {                 // small terms first
        acc = EH_MFMA_BF16(a.l[i], b.h[i], acc);
        acc = EH_MFMA_BF16(a.h[i], b.l[i], acc);
        acc = EH_MFMA_BF16(a.m[i], b.m[i], acc);
        acc = EH_MFMA_BF16(a.m[i], b.h[i], acc);
        acc = EH_MFMA_BF16(a.h[i], b.m[i], acc);
        acc = EH_MFMA_BF16(a.h[i], b.h[i], acc);
    }
}
// two independent products side by side: their matrix instructions alternate, so neither accumulator chain waits for
// its own previous instruction (each accumulator still sees its products in the order of eh_mma64)
__device__ static inline void eh_mma64x2(f32x16& acc0, const eh_frag3& a0, const eh_frag3& b0,
                                         f32x16& acc1, const eh_frag3& a1, const eh_frag3& b1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc0 = EH_MFMA_BF16(a0.l[i], b0.h[i], acc0); acc1 = EH_MFMA_BF16(a1.l[i], b1.h[i], acc1);
        acc0 = EH_MFMA_BF16(a0.h[i], b0.l[i], acc0); acc1 = EH_MFMA_BF16(a1.h[i], b1.l[i], acc1);
        acc0 = EH_MFMA_BF16(a0.m[i], b0.m[i], acc0); acc1 = EH_MFMA_BF16(a1.m[i], b1.m[i], acc1);
        acc0 = EH_MFMA_BF16(a0.m[i], b0.h[i], acc0); acc1 = EH_MFMA_BF16(a1.m[i], b1.h[i], acc1);
        acc0 = EH_MFMA_BF16(a0.h[i], b0.m[i], acc0); acc1 = EH_MFMA_BF16(a1.h[i], b1.m[i], acc1);
        acc0 = EH_MFMA_BF16(a0.h[i], b0.h[i], acc0); acc1 = EH_MFMA_BF16(a1.h[i], b1.h[i], acc1);
    }
}
// half of a 64-deep chunk: the operand groups [I0, I1) only (the loads and splits of the other groups fall away)
template <int I0, int I1>
__device__ static inline void eh_mma64_groups(f32x16& acc, const eh_frag3& a, const eh_frag3& b) {
#pragma unroll
    for (int i = I0; i < I1; ++i) {
        acc = EH_MFMA_BF16(a.l[i], b.h[i], acc);
        acc = EH_MFMA_BF16(a.h[i], b.l[i], acc);
        acc = EH_MFMA_BF16(a.m[i], b.m[i], acc);
        acc = EH_MFMA_BF16(a.m[i], b.h[i], acc);
        acc = EH_MFMA_BF16(a.h[i], b.m[i], acc);
        acc = EH_MFMA_BF16(a.h[i], b.h[i], acc);
    }
}
// ---- fp32 weight tiles split while they are staged (training GEMMs: the weights change every step) -----
// LDS image of a [32 x 64] weight tile as three bf16 planes in FRAGMENT order: dword offset of (row r, k) =
// hh*528 + i*132 + r*4 + (k&7)/2 with hh = k>>5, i = (k&31)>>3.  A lane's operand group i is one 16-byte read
// (conflict-free: the 16 lanes of a b128 group differ in r); the staging thread's four consecutive k are one
// 8-byte write per plane (132 = 4 mod 32 and 528 = 16 mod 32: the 16 (hh, i, half) writers of a row hit all 32 banks).
#define EH_W3_PLANE 1056
typedef unsigned int eh_u32x2 __attribute__((ext_vector_type(2)));
__device__ static inline void eh_stage3_quad(unsigned* __restrict__ tile, int row, int k0, const f32x4& v) {
    unsigned h0, m0, l0, h1, m1, l1;
    eh_split_pair(v[0], v[1], h0, m0, l0);
    eh_split_pair(v[2], v[3], h1, m1, l1);
    unsigned* p = tile + (row >> 5) * 3 * EH_W3_PLANE + (k0 >> 5) * 528 + ((k0 & 31) >> 3) * 132 + (row & 31) * 4 + ((k0 & 7) >> 1);
    *reinterpret_cast<eh_u32x2*>(p) = eh_u32x2{h0, h1};
    *reinterpret_cast<eh_u32x2*>(p + EH_W3_PLANE) = eh_u32x2{m0, m1};
    *reinterpret_cast<eh_u32x2*>(p + 2 * EH_W3_PLANE) = eh_u32x2{l0, l1};
}
__device__ static inline void eh_load_frag3_lds(eh_frag3& f, const unsigned* tile, int j, int r, int hh) {
    const unsigned* p = tile + j * 3 * EH_W3_PLANE + hh * 528 + r * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f.h[i] = *reinterpret_cast<const eh_u32x4*>(p + i * 132);
        f.m[i] = *reinterpret_cast<const eh_u32x4*>(p + EH_W3_PLANE + i * 132);
        f.l[i] = *reinterpret_cast<const eh_u32x4*>(p + 2 * EH_W3_PLANE + i * 132);
    }
}
// split weights in fragment order (k_pack_w): tile (nt, kc) = 3 planes (h, m, l) x [i = 0..3][lane][8 bf16];
// 3072 float-sized words per tile, every load a coalesced 1-KiB row
#define EH_PK_TILE 3072
__device__ static inline void eh_load_frag_packed(eh_frag3& f, const float* __restrict__ packed, int KC, int nt, int kc,
                                                  int lane) {
    const eh_u32x4* p = reinterpret_cast<const eh_u32x4*>(packed + ((size_t)(nt * KC + kc) * EH_PK_TILE)) + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) { f.h[i] = p[i * 64]; f.m[i] = p[256 + i * 64]; f.l[i] = p[512 + i * 64]; }
}
#if EH_BF16X3
typedef eh_frag3 eh_pfrag;                           // operands of a product with packed weights
#define EH_PK_FLOATS(n) ((n) / 2 * 3)                // floats a packed [N][K] matrix occupies
__device__ static inline void eh_to_pfrag(eh_pfrag& o, const eh_frag& f) { eh_split3(o, f); }
#else
typedef eh_frag eh_pfrag;
#define EH_PK_FLOATS(n) (n)
__device__ static inline void eh_to_pfrag(eh_pfrag& o, const eh_frag& f) { o = f; }
#endif

// Generic linear-layer descriptor (see gemm.hip)
struct eh_gemm_desc {
    const float* A; int lda;            // A rows, k < K1
    const int32_t* arow_index;          // optional gather: logical row i reads A row arow_index[i] (packed path)
    const float* A2; int lda2; int K1;  // optional second source for k >= K1 (concat along K); K1 % 64 == 0
    const float* W; int K; int N;       // W [N,K] row-major, K % 64 == 0, N % 32 == 0
    int w_packed;                       // W is in fragment order (eh_load_frag_packed): no LDS staging
    const float* bias;                  // [N] or null
    const float* gbias0;                // [B,ldg] per-graph bias or null
    const float* gbias1;                // [B,ldg] per-graph bias applied where rowflag[row] > 0, or null
    const int32_t* rowflag;
    int ldg;
    float* Y; int ldy; int act;         // act: 0 none, 1 SiLU
    const int32_t* tile_graph;          // node tiling (one graph per tile) or null for plain 32-row tiles
    const int32_t* tile_row0;
    const int32_t* tile_rows;
    int num_tiles;
    int64_t M;                          // rows when tile_row0 == null
    const int32_t* m_dev;               // optional device-side row count overriding M
    int prof_id;                        // kernel class for the optional event timer (prof.h)
    // training extras, honoured by the LDS-staged kernel only (k_gemm): accum 1: Y += result instead of Y = result
    // (fan-in of gradients), 2: float atomics into Y (several products of one grouped launch meet in Y); and a
    // second output Y2 = SiLU(Y) so the pre-activation stays for the backward
    int accum;
    float* Y2;
    const float* dact;                  // result *= SiLU'(dact[row][col]) before it is stored (backward through an activation)
};

// Packed-weight variant: W in fragment order, no LDS, no barriers; every wave streams its fragments
// with coalesced 1-KiB loads (the 4 waves of a block read the same fragments: L1 hits).
template <int NACC>
__device__ static inline void eh_gemm_packed_body(const eh_gemm_desc& d, int ng_per_block, int bx, int by) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int tile = bx * 4 + wave;
    int row0, rows, g = 0;
    if (d.tile_row0) {
        if (tile >= d.num_tiles) return;
        row0 = d.tile_row0[tile]; rows = d.tile_rows[tile]; g = d.tile_graph[tile];
    } else {
        int64_t M = d.m_dev ? (int64_t)(*d.m_dev) : d.M;
        int64_t r0 = (int64_t)tile * 32;
        if (r0 >= M) return;
        row0 = (int)r0; rows = (int)((M - r0) < 32 ? (M - r0) : 32);
    }
    const int r = lane & 31, hh = lane >> 5;
    int64_t arow = row0 + (r < rows ? r : rows - 1);
    if (d.arow_index) arow = d.arow_index[arow];
    const float* ap = d.A + arow * d.lda + hh * 32;
    const float* ap2 = d.A2 ? d.A2 + arow * d.lda2 + hh * 32 : nullptr;
    const int KC = d.K >> 6;
    const int ng_total = d.N / (32 * NACC);
    const int ng0 = by * ng_per_block;
    const int ng1 = (ng0 + ng_per_block < ng_total) ? ng0 + ng_per_block : ng_total;
    const bool single = d.K == 64;
    eh_frag a;
    eh_pfrag a3;
    if (single) { eh_load_frag(a, ap); eh_to_pfrag(a3, a); }
    for (int ng = ng0; ng < ng1; ++ng) {
        f32x16 acc[NACC];
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = eh_zero16();
        for (int kc = 0; kc < KC; ++kc) {
            if (!single) {
                eh_load_frag(a, (kc * 64 < d.K1) ? ap + kc * 64 : ap2 + (kc * 64 - d.K1));
                eh_to_pfrag(a3, a);
            }
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                eh_pfrag b;
                eh_load_frag_packed(b, d.W, KC, ng * NACC + j, kc, lane);
                eh_mma64(acc[j], a3, b);
            }
        }
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
            const int col = (ng * NACC + j) * 32 + r;
            float add0 = d.bias ? d.bias[col] : 0.0f;
            if (d.gbias0) add0 += d.gbias0[(int64_t)g * d.ldg + col];
            const float add1 = d.gbias1 ? d.gbias1[(int64_t)g * d.ldg + col] : 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                int rr = eh_crow(i, lane);
                if (rr < rows) {
                    int64_t grow = row0 + rr;
                    float v = acc[j][i] + add0;
                    if (d.gbias1 && d.rowflag[grow] > 0) v += add1;
                    if (d.act == 1) v = eh_silu(v);
                    d.Y[grow * d.ldy + col] = v;
                }
            }
        }
    }
}


// grid of the packed kernel for a descriptor: blocks along the row tiles (gx), column-group splits, groups per block
void eh_gemm_grid(const eh_gemm_desc& d, int& gx, int& split, int& per, int& nacc);
int eh_launch_gemm(const eh_gemm_desc& d, hipStream_t stream);
// up to EH_GROUP_MAX independent products (N % 64 == 0 each) in one launch
#define EH_GROUP_MAX 8
struct eh_gemm_group { eh_gemm_desc d[EH_GROUP_MAX]; int gx[EH_GROUP_MAX], split[EH_GROUP_MAX], per[EH_GROUP_MAX]; };
int eh_launch_gemm_group(const eh_gemm_desc* ds, int n, hipStream_t stream);
