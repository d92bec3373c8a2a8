// Fused multi-head attention core for short sequences (ViT-B/16: T = 197, text: T = 77 causal), head_dim 64.
//
// attn_rows_kernel<T, NKT>: one 256-thread workgroup per (batch, head); the head's K and V ([T,64] each, read
// straight out of the packed QKV activation: 128-byte rows) are staged once into LDS and shared by the 4 waves,
// each of which owns 32-query blocks.  Per block the wave computes S^T = K.Q^T with v_mfma_f32_32x32x16 (keys on
// the accumulator ROWS, the query on the LANE), so a query's whole score row (NKT*32 keys) sits in two lanes
// (l, l+32): softmax needs one cross-lane exchange, no LDS, and - since T <= 224 - no online rescaling.
// The exponentiated accumulators are converted to 16-bit and fed back as the B operand of O^T = V^T.P^T (the
// "accumulator tile as next MFMA's operand" identity: element j of lane half h is key 16s + 8(j>>2) + 4h + (j&3)),
// with V^T fragments produced by the transposing LDS read ds_read_b64_tr_b16 from the row-major V image.
// LDS swizzles: K rows are read with ds_read_b128 (chunk ^= (row>>1)&7), V rows with the tr read (64-B half ^= (row>>1)&1).
#include "leclip_common.h"
#include <stdlib.h>


namespace {

struct AttnArgs {
    const void* qkv;
    void* out;
    int T, heads;
    int64_t ld_qkv, ld_out;
    float scale_log2e;  // scale * log2(e)
    int causal;
#ifdef LECLIP_DIAG
    WgLog wglog;
#endif
    int reverse;        // attn_heads_kernel: walk the (batch, head) pairs from the last to the first
    int q_rows;         // 0: every query row; n > 0: only the first n query rows of every (batch, head) are computed and stored
                        // (leclip_attention_prefix_fwd: the last block of the image tower needs the class token's row only)
};

constexpr int STREAM_TMAX = 640;
template <int N> struct IntC { static constexpr int value = N; };
typedef __attribute__((ext_vector_type(4))) int attn_i32x4;
typedef __attribute__((ext_vector_type(2))) int attn_i32x2;

__device__ __forceinline__ float lane32_max(float v) {   // max over the lane pair (l, l ^ 32), in both lanes
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float lane32_sum(float v) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}


// Output block of one 32-query block: lane (r = lane & 31, h = lane >> 5) holds O[q0 + r][32 i + 8 g + 4 h + e] in o[i][4 g + e].  The two
// lanes of a query exchange halves (v_permlane32_swap, no LDS) so that each ends up with 8 consecutive columns: four 16-byte stores
// per lane and block instead of eight 8-byte ones - the store tail of these kernels is bound by store ISSUE, not by bytes.
template <typename T>
__device__ __forceinline__ void attn_store_block(const f32x16 (&o)[2], float inv, T* op, int fh, bool on) {
    typedef typename VecOf<T>::v4 v4;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            v4 wa, wb;
#pragma unroll
            for (int e = 0; e < 4; ++e) { wa[e] = (T)(o[i][8 * k + e] * inv); wb[e] = (T)(o[i][8 * k + 4 + e] * inv); }
            const u32x2 a2 = __builtin_bit_cast(u32x2, wa), b2 = __builtin_bit_cast(u32x2, wb);
            // swap(X = a, Y = b): X[32..63] <-> Y[0..31].  h = 0: (own a, partner's a) = columns +0..7; h = 1: (partner's b, own b) = +8..15
            const u32x2 s0 = __builtin_amdgcn_permlane32_swap(a2[0], b2[0], false, false);
            const u32x2 s1 = __builtin_amdgcn_permlane32_swap(a2[1], b2[1], false, false);
            attn_i32x4 w;
            w[0] = (int)s0[0]; w[1] = (int)s1[0]; w[2] = (int)s0[1]; w[3] = (int)s1[1];
            if (on) *(attn_i32x4*)(op + 32 * i + 16 * k + 8 * fh) = w;
        }
}

// One 32-query block of one (batch, head): S^T = K.Q^T, masked softmax over keys, O^T = V^T.P^T, store.
// sK / sV: the head's K and V images in LDS (swizzled as described above); `base` points at q[b, 0, head, 0].
// Q fragments of a 32-query block (MFMA B operand): lane (r = lane&31, h = lane>>5) holds Q[q0 + r][16s + 8h + j].
template <typename T>
__device__ __forceinline__ void attn_load_q(const AttnArgs& a, const T* base, int qb, int lane,
                                            typename VecOf<T>::v8 (&qf)[4]) {
    typedef typename VecOf<T>::v8 v8;
    const int qi = qb * 32 + (lane & 31);
    const int qrow = qi < a.T ? qi : a.T - 1;
    const T* qp = base + (int64_t)qrow * a.ld_qkv + (lane >> 5) * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const v8*)(qp + s * 16);
}

// STORE_ALL: lanes whose query index is past T store the (identical) result of the clamped row T-1 instead of being
// masked off, so every wave issues the same number of store instructions (counted vmcnt waits in the pipelined kernel).
// LIVE: how many of the LAST key tile's four 8-key groups can hold a valid key (T - 32 (NKT - 1) <= 8 LIVE; 4 = no assumption).  A lane's registers
// 4 g .. 4 g + 3 of a score tile are the keys of group g, so a dead group's mask, maximum, exponentials and conversions are not issued at all
// and - LIVE <= 2 - neither is the tile's second P.V step (its probabilities are exact zeros: the result is the same bits).  ViT-B/16: T = 197,
// five keys in the seventh tile, LIVE = 1: 12 of a lane's 112 exponentials and 2 of 28 P.V MFMAs per query block.
template <typename T, int NKT, bool STORE_ALL, int LIVE = 4>
__device__ __forceinline__ void attn_qblock(const AttnArgs& a, const char* sK, const char* sV,
                                            const typename VecOf<T>::v8 (&qf)[4], T* obase, int qb, int lane) {
    typedef typename VecOf<T>::v8 v8;
    typedef typename VecOf<T>::v4 v4;
    const int fr = lane & 31, fh = lane >> 5;
    const int li = lane & 15, dgrp = (lane >> 4) & 1;
    const char* kbase[4];
    const char* vbase[2];
#pragma unroll
    for (int s = 0; s < 4; ++s) kbase[s] = sK + fr * 128 + (((2 * s + fh) ^ ((fr >> 1) & 7)) << 4);
#pragma unroll
    for (int i = 0; i < 2; ++i)
        vbase[i] = sV + (4 * fh + (li >> 2)) * 128 + ((64 * i) ^ (((li >> 3) & 1) << 6)) + 32 * dgrp + 8 * (li & 3);
    {
        const int qi = qb * 32 + fr;
        const int qrow = qi < a.T ? qi : a.T - 1;
        f32x16 sc[NKT];
        if constexpr (STORE_ALL) {
            // pipelined kernel: K fragments one key tile (4 reads) ahead of their MFMAs - the wave's S -> softmax -> O chain,
            // not HBM, is what bounds that kernel, so its LDS round trips are taken off the chain
            // Written as plain loads (round 2: "one key tile ahead") hipcc serialised this phase: every MFMA behind an
            // s_waitcnt lgkmcnt(0) for its own, just issued ds_read_b128 - one exposed LDS round trip per MFMA, 28 per head, the
            // matrix pipe idle three quarters of the phase.  The reads are inline asm (invisible to the scheduler, which cannot
            // then fold the ring back into one register set), a ring of KR fragments, KR - 1 reads ahead of the MFMA that
            // consumes them; the counted wait carries the fragment as an in/out operand, so the MFMA cannot move above it.
            typedef __attribute__((ext_vector_type(4))) int i32x4;
            constexpr int KR = 4, NST = 4 * NKT;
            unsigned ka[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) ka[s] = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)kbase[s];
            i32x4 kr[KR];
#define K_ISSUE(step) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kr[(step) % KR]) : "v"(ka[(step) & 3]), "n"(((step) >> 2) * 4096))
#pragma unroll
            for (int st = 0; st < KR - 1; ++st) K_ISSUE(st);
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                const int kt = st >> 2, s = st & 3;
                if (st + KR - 1 < NST) K_ISSUE(st + KR - 1);
                // DS operations return in order: all but the reads of the younger steps have arrived
                if (st + KR - 1 < NST) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(kr[st % KR]) : "n"(KR - 1));
                else if (st + 2 < NST) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(kr[st % KR]));
                else if (st + 1 < NST) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(kr[st % KR]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kr[st % KR]));
                if (s == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[kt][r] = 0.f;
                }
                sc[kt] = mfma_32x32x16(__builtin_bit_cast(v8, kr[st % KR]), qf[s], sc[kt]);
                __builtin_amdgcn_sched_barrier(0);
            }
#undef K_ISSUE
        } else {
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[kt][r] = 0.f;
            // key row kt*32 + fr: the swizzle term ((row>>1)&7) depends on fr only, so each k-step has one
            // lane address and the tile index is an immediate offset
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const v8 kf = *(const v8*)(kbase[s] + kt * 4096);
                sc[kt] = mfma_32x32x16(kf, qf[s], sc[kt]);
            }
        }
        }
        // ---- mask + softmax over keys (rows of S^T); this lane holds keys kt*32 + (r&3) + 8(r>>2) + 4h
        const int klimit = a.causal ? (qrow < a.T - 1 ? qrow : a.T - 1) : a.T - 1;  // last valid key
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            // a key tile needs masking only if it reaches past the last valid key (wave-uniform test): for the
            // unmasked image tower that is the final tile alone
            const bool partial = a.causal || (kt * 32 + 31 > a.T - 1);
            const int nr = kt == NKT - 1 ? 4 * LIVE : 16;   // registers of this tile that can hold a valid key
            if (partial) {
#pragma unroll
                for (int r = 0; r < nr; ++r) {
                    const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    sc[kt][r] = key <= klimit ? sc[kt][r] : -3.0e38f;
                }
            }
#pragma unroll
            for (int r = 0; r < nr; ++r) mx = fmaxf(mx, sc[kt][r]);
        }
        mx = lane32_max(mx);   // the query's other lane (v_permlane32_swap: no LDS round trip)
        const float mb = mx * a.scale_log2e;
        // Two keys at a time as a float pair: the scale-and-shift and the row sum issue as packed fp32 instructions (v_pk_fma_f32 /
        // v_pk_add_f32: half the issue slots of that half of the softmax; the sum runs as an even-key and an odd-key partial).
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 sum2 = {0.f, 0.f};
        const f2 scl2 = (f2)(a.scale_log2e), nmb2 = (f2)(-mb);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                if (kt == NKT - 1 && r >= 4 * LIVE) {   // a dead group: what the masked path computes for it, without computing it
                    sc[kt][r] = 0.f;
                    sc[kt][r + 1] = 0.f;
                    continue;
                }
                // v_exp_f32 directly: arguments are <= 0, results below 2^-126 flush to 0 (masked keys: exactly 0)
                f2 x = {sc[kt][r], sc[kt][r + 1]};
                x = __builtin_elementwise_fma(x, scl2, nmb2);
                const f2 p = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
                sc[kt][r] = p.x;
                sc[kt][r + 1] = p.y;
                sum2 += p;
            }
        const float sum = lane32_sum(sum2.x + sum2.y);
        const float inv = 1.0f / sum;

        // ---- O^T[d][q] = sum_key V^T[d][key] P^T[key][q]
        f32x16 o[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
        if constexpr (!STORE_ALL) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    v8 pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (T)sc[kt][8 * s2 + j];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        v8 vf;
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            // key = kt*32 + 16*s2 + 8u + 4h + (li>>2), d0 = 32i + 16*dgrp + 4*(li&3); the 64-byte swizzle
                            // bit ((key>>1)&1) is (li>>3)&1: lane-constant, folded into vbase[i]
                            const v4 t4 = lds_read_tr16((const T*)(vbase[i] + (kt * 32 + 16 * s2 + 8 * u) * 128));
#pragma unroll
                            for (int e = 0; e < 4; ++e) vf[4 * u + e] = t4[e];
                        }
                        o[i] = mfma_32x32x16(vf, pf, o[i]);
                    }
                }
            }
        } else {
            // Pipelined kernel: LDS-DMA for the next head is in flight, and hipcc puts s_waitcnt vmcnt(0) in front of the
            // ds_read_b64_tr_b16 *builtin* (it cannot see that the read does not alias the DMA target).  The same
            // instruction through inline asm is invisible to that pass; its completion is then ours to wait for
            // (lgkmcnt(0) + sched_barrier before the consuming MFMAs), software-pipelined one step ahead.
            typedef __attribute__((ext_vector_type(2))) int i32x2;
            const unsigned va0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)vbase[0];
            const unsigned va1 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)vbase[1];
            i32x2 vr[3][4];   // [step % 3][i*2 + u]: transposing reads two steps ahead of their MFMAs
#define TR_ISSUE(slot, step)                                                                                         \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int u = 0; u < 2; ++u)               \
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2"                                                    \
                             : "=v"(vr[slot][i * 2 + u])                                                              \
                             : "v"(i ? va1 : va0), "n"((((step) >> 1) * 32 + 16 * ((step) & 1) + 8 * u) * 128));
            constexpr int NPV = 2 * NKT - (LIVE <= 2 ? 1 : 0);   // 16-key P.V steps: the last tile's second step is all zeros when its groups 2, 3 are dead
            static_assert(NPV >= 2, "ring primes two steps");
            TR_ISSUE(0, 0)
            TR_ISSUE(1, 1)
#pragma unroll
            for (int st = 0; st < NPV; ++st) {
                const int kt = st >> 1, s2 = st & 1, slot = st % 3;
                if (st + 2 < NPV) {
                    if (slot == 0) { TR_ISSUE(2, st + 2) } else if (slot == 1) { TR_ISSUE(0, st + 2) } else { TR_ISSUE(1, st + 2) }
                }
                v8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (T)sc[kt][8 * s2 + j];
                // DS operations return in order: everything but the 4 reads of each younger step has arrived
                if (st + 2 < NPV) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                else if (st + 1 < NPV) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                v8 vf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    typedef __attribute__((ext_vector_type(4))) int i32x4;
                    i32x4 w;
                    w[0] = vr[slot][i * 2][0]; w[1] = vr[slot][i * 2][1]; w[2] = vr[slot][i * 2 + 1][0]; w[3] = vr[slot][i * 2 + 1][1];
                    vf[i] = __builtin_bit_cast(v8, w);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) o[i] = mfma_32x32x16(vf[i], pf, o[i]);
            }
#undef TR_ISSUE
        }
        // ---- store: lane owns query qi
        attn_store_block<T>(o, inv, obase + (int64_t)(STORE_ALL ? qrow : qi) * a.ld_out, fh, STORE_ALL || qi < a.T);
    }
}

template <typename T, int NKT>
__global__ __launch_bounds__(256, 2) void attn_rows_kernel(AttnArgs a) {
    typedef typename VecOf<T>::v8 v8;
    constexpr int TP = NKT * 32;
    __shared__ __attribute__((aligned(16))) char sK[TP * 128];
    __shared__ __attribute__((aligned(16))) char sV[TP * 128];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / a.heads, h = blockIdx.x - b * a.heads;
    const int d_model = a.heads * 64;
    const T* base = (const T*)a.qkv + (int64_t)b * a.T * a.ld_qkv + h * 64;

    // ---- stage K and V: 8 threads x 16 B per 128-byte row, 32 rows per pass; all 2*NKT loads of a thread are issued before the
    // first LDS write (with one pass in flight at a time the prefix call, which computes a single query block, was bound by this loop)
    {
        const int c = tid & 7;
        v8 kv[NKT], vv[NKT];
#pragma unroll
        for (int u = 0; u < NKT; ++u) {
            const int r = (tid >> 3) + 32 * u;
            const int gr = r < a.T ? r : a.T - 1;   // (rows past T are zeroed below; the clamped load keeps the loop branch-free)
            const T* row = base + (int64_t)gr * a.ld_qkv;
            kv[u] = *(const v8*)(row + d_model + c * 8);
            vv[u] = *(const v8*)(row + 2 * d_model + c * 8);
        }
#pragma unroll
        for (int u = 0; u < NKT; ++u) {
            const int r = (tid >> 3) + 32 * u;
            if (r >= a.T) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { kv[u][i] = (T)0.f; vv[u][i] = (T)0.f; }
            }
            *(v8*)(sK + r * 128 + ((c ^ ((r >> 1) & 7)) << 4)) = kv[u];
            *(v8*)(sV + r * 128 + ((c ^ (((r >> 1) & 1) << 2)) << 4)) = vv[u];
        }
    }
    __syncthreads();

    const int nqb = ((a.q_rows > 0 ? a.q_rows : a.T) + 31) >> 5;   // (a block is computed whole: with q_rows = 1, rows 0..31 are stored)
    T* obase = (T*)a.out + (int64_t)b * a.T * a.ld_out + h * 64;
    for (int qb = wave; qb < nqb; qb += 4) {
        v8 qf[4];
        attn_load_q<T>(a, base, qb, lane, qf);
        attn_qblock<T, NKT, false>(a, sK, sV, qf, obase, qb, lane);
    }
}

// ---------------------------------------------------------------- pipelined kernel for many heads (ViT at large batch)
// 512 threads = 8 waves, one persistent workgroup per CU walking (batch, head) pairs.  K/V of head i+1 stream into the
// second LDS buffer by LDS-DMA (global_load_lds_dwordx4; the swizzles are applied to the per-lane SOURCE address, rows past
// T re-read row T-1: masked keys / zero probabilities make them inert) and its Q fragments into registers while head i
// is computed; wave w owns query block w.  The wait is counted: every wave ends a head with exactly 4 output stores,
// so "all but the newest 4" retires its share of the next head's K/V and Q without draining those stores.  One barrier
// per head.
template <typename T, int NKT, int LIVE>
__global__ __launch_bounds__(512, 2) void attn_heads_kernel(AttnArgs a, int total_heads) {
    typedef typename VecOf<T>::v8 v8;
    constexpr int TP = NKT * 32, BUF = TP * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][K | V][BUF] + [8 waves][4 KiB Q]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d_model = a.heads * 64;
    const int last = total_heads - 1;

    // Walk order: the LAST (batch, head) pairs first.  The packed qkv rows were written by the GEMM in front of this kernel in ascending row
    // order, so the tail of the tensor is what the memory-side cache (256 MiB; the tensor is 232 MB at B = 256) still holds - an
    // ascending walk asks for the oldest lines first and evicts the youngest as it goes (round 4: profiles/r04_walk_order.txt).
    auto eff = [&](int hd) { return a.reverse ? total_heads - 1 - hd : hd; };
    // A head's source as a buffer descriptor built on the scalar unit (base = q[b, 0, head, 0], size = the rest of the image's rows), so that the
    // per-lane part of every piece's address is a 32-bit byte offset that does not change from head to head: the loop's LDS-DMA issue costs no
    // vector address arithmetic (as global_load_lds with 64-bit lane addresses it was 33 vector instructions per head and wave).
    auto head_rsrc = [&](int hd) {
        hd = eff(hd);
        const int b = hd / a.heads, h = hd - b * a.heads;
        const char* base = (const char*)((const T*)a.qkv + (int64_t)b * a.T * a.ld_qkv + h * 64);
        const unsigned long long v = (unsigned long long)base;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        const unsigned size = __builtin_amdgcn_readfirstlane((unsigned)((int64_t)a.T * a.ld_qkv * 2 - h * 128));
        return __builtin_amdgcn_make_buffer_rsrc((char*)(((unsigned long long)hi << 32) | lo), 0, (int)size, 0x00020000);
    };
    // K then V, TP/8 pieces of 8 rows x 128 B each; wave w takes pieces w, w+8, ... (NKT per wave)
    unsigned kv_off[NKT], q_off[4];
    int kv_dst[NKT];
#pragma unroll
    for (int u = 0; u < NKT; ++u) {
        const int pc = wave + 8 * u;
        const int isv = pc >= TP / 8 ? 1 : 0;
        const int piece = pc - isv * (TP / 8);
        const int row = piece * 8 + (lane >> 3);
        const int p = lane & 7;
        const int c = isv ? (p ^ (((row >> 1) & 1) << 2)) : (p ^ ((row >> 1) & 7));
        const int grow = row < a.T ? row : a.T - 1;
        kv_off[u] = (unsigned)grow * (unsigned)a.ld_qkv * 2u + (unsigned)((1 + isv) * d_model * 2 + c * 16);
        kv_dst[u] = isv * BUF + piece * 1024;
    }
    auto issue = [&](__amdgpu_buffer_rsrc_t r, int buf) {
#pragma unroll
        for (int u = 0; u < NKT; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(smem + buf * 2 * BUF + kv_dst[u]), 16, (int)kv_off[u], 0, 0, 0);
    };

    // Q of the wave's own 32-query block also travels by LDS-DMA into a wave-private 4 KiB image (K-style swizzle), so the
    // loop holds no VGPR-destination global load: hipcc then inserts no vmcnt waits of its own and the counted waits below
    // are the only ones.
    char* sQ = smem + 4 * BUF + wave * 4096;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int row = u * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int qi = wave * 32 + row;
        const int grow = qi < a.T ? qi : a.T - 1;
        q_off[u] = (unsigned)grow * (unsigned)a.ld_qkv * 2u + (unsigned)(c * 16);
    }
    auto issue_q = [&](__amdgpu_buffer_rsrc_t r) {
#pragma unroll
        for (int u = 0; u < 4; ++u) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(sQ + u * 1024), 16, (int)q_off[u], 0, 0, 0);
    };
    const char* qrd = sQ + (lane & 31) * 128;
    const int qsw = ((lane & 31) >> 1) & 7, qh = lane >> 5;

#ifdef LECLIP_DIAG
    unsigned long long wl_t0 = 0, wl_c0 = 0;
    if (a.wglog.buf && tid == 0) { wl_t0 = __builtin_amdgcn_s_memrealtime(); wl_c0 = __builtin_amdgcn_s_memtime(); }
#endif
    int hd = blockIdx.x;
    const bool active = wave * 32 < a.T;   // wave-uniform
#ifdef LECLIP_ATTN_YOUNG_PRIO   // A/B builds: static priority for the second-dispatched half (MI355X_MICROARCH.md, two waves per SIMD, item 4)
    if (wave >= 4) __builtin_amdgcn_s_setprio(LECLIP_ATTN_YOUNG_PRIO);
#endif
    {
        const __amdgpu_buffer_rsrc_t r0 = head_rsrc(hd);
        issue(r0, 0);
        if (active) issue_q(r0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    for (int it = 0; hd < total_heads; ++it) {
        const int cur = it & 1;
        const int nxt = hd + gridDim.x;
        const int nxt_c = nxt < total_heads ? nxt : last;   // past the end: a harmless re-load keeps the op counts uniform
        // One barrier per head: every wave has (a) finished head it-1, so buffer cur^1 may be overwritten, and
        // (b) passed the counted wait at the end of its previous iteration, so every share of head `hd` is in buffer cur.
        __builtin_amdgcn_s_barrier();
#ifdef LECLIP_ATTN_SKEW   // A/B builds: waves 4..7 start each head LECLIP_ATTN_SKEW x 64 cycles late, so that a SIMD's two waves are a phase apart
        if (wave >= 4) __builtin_amdgcn_s_sleep(LECLIP_ATTN_SKEW);
#endif
        const __amdgpu_buffer_rsrc_t nb = head_rsrc(nxt_c);
        if (active) {
            v8 q[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) q[s] = *(const v8*)(qrd + (((2 * s + qh) ^ qsw) << 4));
            __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0), as the compiler's own instruction (its scoreboard then knows the Q fragments are in: no second wait in front of the first MFMA): the Q image may be refilled
            issue(nb, cur ^ 1);
            issue_q(nb);
            const int he = eff(hd);
            const int b = he / a.heads, h = he - b * a.heads;
            T* obase = (T*)a.out + (int64_t)b * a.T * a.ld_out + h * 64;
            attn_qblock<T, NKT, true, LIVE>(a, smem + cur * 2 * BUF, smem + cur * 2 * BUF + BUF, q, obase, wave, lane);
            // Everything older than this head's 4 output stores has completed: the next head's K/V share and Q image
            // (issued a whole head ago) are in, without draining the stores just issued.
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            // a wave without a query block (wave 7 when T <= 224) only moves its share of K/V
            issue(nb, cur ^ 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        hd = nxt;
    }
#ifdef LECLIP_DIAG
    if (a.wglog.buf && tid == 0) wglog_end(a.wglog, 0x200u, wl_t0, wl_c0);
#endif
}

// ---------------------------------------------------------------- streaming kernel for long sequences (ViT-L/14@336: T = 577)
// One 512-thread workgroup per (batch, head); the head's whole K and V (T <= 640: 2 x 80 KiB) sit in LDS, each wave owns 32-query
// blocks and walks the keys in chunks of 4 tiles (128 keys) with the online-softmax recurrence: the running row maximum m lives in
// the two lanes that hold a query, O^T accumulators are rescaled by exp2(scale*(m_old - m_new)) when the maximum moves.  Same MFMA
// formulation as attn_qblock (S^T = K.Q^T, P fed back as the B operand, V^T by ds_read_b64_tr_b16).
// Round 3 wrote the per-chunk chain out by hand: round 2's plain source compiled to ~600 vector
// instructions per chunk and wave - 64 zero-initialising moves (tiles that might be skipped), 64 moves merging the masked and the
// unmasked path, two ds_bpermute round trips for the row maximum and sum - behind LDS reads that each MFMA waited for.  Here:
//   * chunks are templates on (tiles, mask mode): the full chunks run branch-free, only the last tile of the last chunk is masked;
//   * K fragments and V^T fragments come from inline-asm reads in rings (3 / 2 steps ahead of their MFMAs), the first V^T
//     fragments are requested before the softmax arithmetic; counted lgkmcnt waits tied to the registers they release;
//   * the row maximum crosses the two lanes of a query with v_permlane32_swap (no LDS); the row SUM stays a per-lane partial
//     for the whole query block (both lanes scale it by the same factors) and is combined once, at the end;
//   * scale / subtract, the sum and the rescaling of O are packed fp32 operations (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32).
// One chunk of NT key tiles (32 keys each) against the wave's 32-query block.  ka[s] / va0 / va1: this lane's LDS byte addresses
// of the chunk's first K fragment row (k-step s) and V^T blocks; lim = (last valid key) - (chunk's first key) - 4 * (lane >> 5).
// MASK: 0 none, 1 the last tile only, 2 every tile (causal).  FIRST: the block's first chunk.
// The reference maximum is LAZY (round 3, third step; profiles/r03_attention_lazy.txt): the kernel's time follows its instruction count, so
// this form removes vector work instead of moving it:
//   * Q is pre-multiplied by scale * log2(e) once per query block, so a score is already an exp2 argument;
//   * the subtraction of the row's reference maximum is done BY THE MFMA: the S accumulators start from -m_ref (`negm`, 16 registers that
//     change only when m_ref does) instead of 0, the matrix pipe delivers S' = S - m_ref;
//   * m_ref is the maximum of the FIRST chunk (accumulators from 0, m_ref := chunk maximum) and moves only when a later chunk exceeds it
//     by more than LAZY_THR (2^8: probabilities stay <= 256, exact in fp32 sums, far inside fp16's range) for some query of the wave - a
//     wave-uniform, rarely taken branch that rescales O and the row sum; every other chunk is max3 -> exp2 -> add -> convert, with no
//     fma and no rescale of O (190 vector instructions per chunk against 227).
// The result is softmax(S) V all the same: numerator and denominator carry the same factor 2^(m_true - m_ref).  kernel_check's and
// tests/test_gpu_parity.py's spiked cases put maxima into later chunks, above and below the threshold (guide rule 26).
constexpr float LAZY_THR = 8.0f;
#if defined(LECLIP_ATTN_PRIO_M)     // experiment: the wave in its matrix phases outranks its SIMD partner
#define LAZY_PRIO_M() __builtin_amdgcn_s_setprio(1)
#define LAZY_PRIO_V() __builtin_amdgcn_s_setprio(0)
#elif defined(LECLIP_ATTN_PRIO_V)   // experiment: the wave in its vector phase outranks its SIMD partner
#define LAZY_PRIO_M() __builtin_amdgcn_s_setprio(0)
#define LAZY_PRIO_V() __builtin_amdgcn_s_setprio(1)
#else
#define LAZY_PRIO_M() do { } while (0)
#define LAZY_PRIO_V() do { } while (0)
#endif
template <typename T, int NT, int MASK, bool FIRST>
__device__ __forceinline__ void stream_chunk_lazy(const unsigned (&ka)[4], unsigned va0, unsigned va1, const typename VecOf<T>::v8 (&qf)[4],
                                                  f32x16 (&o)[2], f32x16& negm, float& m_ref, float& l_part, int lim) {
    typedef typename VecOf<T>::v8 v8;
    constexpr int KR = 8, KA = KR - 1, VR = 4, VA = VR - 1, NST = 4 * NT, NPV = 2 * NT;
    f32x16 sc[NT];
    attn_i32x4 kr[KR];
    attn_i32x2 vr[VR][4];   // [PV step % VR][d block * 2 + key half]
#define SK_ISSUE(step) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kr[(step) % KR]) : "v"(ka[(step) & 3]), "n"(((step) >> 2) * 4096))
#define SV_ISSUE(step)                                                                                                          \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) _Pragma("unroll") for (int u_ = 0; u_ < 2; ++u_)                            \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vr[(step) % VR][i_ * 2 + u_]) : "v"(i_ ? va1 : va0),           \
                     "n"((((step) >> 1) * 32 + 16 * ((step) & 1) + 8 * u_) * 128))
#define SK_WAIT(n, reg) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(reg) : "n"(n))
    // ---- S' = K . Q'^T - m_ref
    LAZY_PRIO_M();
#pragma unroll
    for (int st = 0; st < KA && st < NST; ++st) SK_ISSUE(st);
#pragma unroll
    for (int st = 0; st < NST; ++st) {
        const int kt = st >> 2, s = st & 3;
        if (st + KA < NST) SK_ISSUE(st + KA);
        switch (st + KA < NST ? KA : NST - 1 - st) {   // (compile-time after unrolling)
            case 0: SK_WAIT(0, kr[st % KR]); break;
            case 1: SK_WAIT(1, kr[st % KR]); break;
            case 2: SK_WAIT(2, kr[st % KR]); break;
            case 3: SK_WAIT(3, kr[st % KR]); break;
            case 4: SK_WAIT(4, kr[st % KR]); break;
            case 5: SK_WAIT(5, kr[st % KR]); break;
            case 6: SK_WAIT(6, kr[st % KR]); break;
            default: SK_WAIT(7, kr[st % KR]); break;
        }
        static_assert(KA <= 7, "wait table");
        const v8 kf = __builtin_bit_cast(v8, kr[st % KR]);
        if (s == 0) {
            if constexpr (FIRST) {
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                sc[kt] = mfma_32x32x16(kf, qf[s], z);
            } else {
                sc[kt] = mfma_32x32x16(kf, qf[s], negm);
            }
        } else {
            sc[kt] = mfma_32x32x16(kf, qf[s], sc[kt]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int st = 0; st < VA && st < NPV; ++st) SV_ISSUE(st);
    // ---- mask, chunk maximum (relative to m_ref unless FIRST)
    LAZY_PRIO_V();
    float mx = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        if (MASK == 2 || (MASK == 1 && kt == NT - 1)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[kt][r] = lim >= kt * 32 + (r & 3) + 8 * (r >> 2) ? sc[kt][r] : -3.0e38f;
        }
#pragma unroll
        for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, sc[kt][r]), sc[kt][r + 1]);   // (v_max3_f32)
    }
    mx = lane32_max(mx);
    if constexpr (FIRST) {
        m_ref = mx;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[kt][r] -= mx;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] = -mx;
    } else if (__builtin_amdgcn_ballot_w64(mx > LAZY_THR) != 0) {   // rare: some query's maximum moved by more than 2^LAZY_THR
        const float delta = fmaxf(mx, 0.f);
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        m_ref += delta;
        l_part *= alpha;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[kt][r] -= delta;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] = -m_ref;
    }
    float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = __builtin_amdgcn_exp2f(sc[kt][r]);   // masked keys: exp2(-huge) = exactly 0
            sc[kt][r] = pv;
            ps[r & 3] += pv;
        }
    l_part += (ps[0] + ps[1]) + (ps[2] + ps[3]);
    LAZY_PRIO_M();
    // ---- O^T += V^T . P^T
#pragma unroll
    for (int st = 0; st < NPV; ++st) {
        const int kt = st >> 1, s2 = st & 1, slot = st % VR;
        if (st + VA < NPV) SV_ISSUE(st + VA);
        v8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (T)sc[kt][8 * s2 + j];
#define SV_WAIT(n) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(vr[slot][0]), "+v"(vr[slot][1]), "+v"(vr[slot][2]), "+v"(vr[slot][3]) : "n"(n))
        switch (st + VA < NPV ? VA : NPV - 1 - st) {
            case 0: SV_WAIT(0); break;
            case 1: SV_WAIT(4); break;
            case 2: SV_WAIT(8); break;
            default: SV_WAIT(12); break;
        }
        static_assert(VA <= 3, "wait table");
#undef SV_WAIT
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            attn_i32x4 w;
            w[0] = vr[slot][i * 2][0]; w[1] = vr[slot][i * 2][1]; w[2] = vr[slot][i * 2 + 1][0]; w[3] = vr[slot][i * 2 + 1][1];
            o[i] = mfma_32x32x16(__builtin_bit_cast(v8, w), pf, o[i]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#undef SK_ISSUE
#undef SV_ISSUE
#undef SK_WAIT
}

#ifdef LECLIP_ATTN_STAMPS   // diagnostic variant builds only: shader-clock stamps of waves 0 and 4 of the first 2048 workgroups
__device__ unsigned long long g_attn_stamps[2048 * 2 * 16];
#define ASTAMP(k)                                                                                                   \
    do {                                                                                                            \
        if ((wave & 3) == 0 && lane == 0 && blockIdx.x < 2048)                                                      \
            g_attn_stamps[(blockIdx.x * 2 + (wave >> 2)) * 16 + (k)] = __builtin_amdgcn_s_memtime();                \
    } while (0)
#define ASTAMP_RT(k)                                                                                                \
    do {                                                                                                            \
        if ((wave & 3) == 0 && lane == 0 && blockIdx.x < 2048)                                                      \
            g_attn_stamps[(blockIdx.x * 2 + (wave >> 2)) * 16 + (k)] = __builtin_amdgcn_s_memrealtime();            \
    } while (0)
#else
#define ASTAMP(k) do { } while (0)
#define ASTAMP_RT(k) do { } while (0)
#endif

template <typename T, bool CAUSAL>
__global__ __launch_bounds__(512, 2) void attn_stream_kernel(AttnArgs a, int TP) {
    typedef typename VecOf<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // K [TP][128 B] | V [TP][128 B]
    char* sK = smem;
    char* sV = smem + TP * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / a.heads, h = blockIdx.x - b * a.heads;
    const int d_model = a.heads * 64;
    const T* base = (const T*)a.qkv + (int64_t)b * a.T * a.ld_qkv + h * 64;
    ASTAMP(0);
    ASTAMP_RT(14);
    // K and V of the head by LDS-DMA, every piece (8 rows x 128 B) in flight at once, then the wave's first query block (below).  (Starting chunk c of the first query block as soon as the pieces of chunks 0 .. c have landed - pieces issued in
    // chunk order, a counted vmcnt and a barrier per chunk - was built and measured 5 % SLOWER: a CU takes the 148 KB of a head in
    // at 13 bytes per cycle whatever the order, the first chunk's pieces land almost as late as the last one's, and the loads then
    // run beside the first block's arithmetic instead of in front of it; profiles/r03_attention_stream.txt.)
    const int nqb = ((a.q_rows > 0 ? a.q_rows : a.T) + 31) >> 5;
    const int nch = TP >> 7;
    attn_i32x4 q0[4];
    for (int ch = 0; ch < nch; ++ch) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int isv = u >> 1;
            const int piece = ch * 16 + wave + 8 * (u & 1);
            const int row = piece * 8 + (lane >> 3);
            const int p = lane & 7;
            const int cc = isv ? (p ^ (((row >> 1) & 1) << 2)) : (p ^ ((row >> 1) & 7));
            const int grow = row < a.T ? row : a.T - 1;
            const T* src = base + (int64_t)grow * a.ld_qkv + (1 + isv) * d_model + cc * 8;
            __builtin_amdgcn_global_load_lds((const void*)src, LDS_PTR((isv ? sV : sK) + piece * 1024), 16, 0, 0);
        }
    }
    // The wave's first Q block: loads AND their wait in ONE asm statement (early-clobber outputs), behind the LDS-DMA issue loop.  Inline
    // asm so that the compiler has no pending load on these registers at the head of the query-block loop (it would put s_waitcnt
    // vmcnt(0) in front of every block's first MFMA, where it waits for the NEXT block's Q, just requested); ONE statement so that no
    // compiler-scheduled instruction can sit between the issue and the wait - an asm load's destination counts as written when its
    // statement ends, so across compiler code it may be copied, spilled or re-used before the data lands (round 3 issued these loads
    // in front of the DMA loop and waited behind it: tests/isa_audit.py now checks the shipped code objects for that pattern).  vmcnt
    // retires in order, so the wait also covers every K / V piece issued above.
    {
        const int qi = wave * 32 + (lane & 31);
        const int qrow = qi < a.T ? qi : a.T - 1;
        const T* qp = base + (int64_t)qrow * a.ld_qkv + (lane >> 5) * 8;
        asm volatile("global_load_dwordx4 %0, %4, off\n\t"
                     "global_load_dwordx4 %1, %4, off offset:32\n\t"
                     "global_load_dwordx4 %2, %4, off offset:64\n\t"
                     "global_load_dwordx4 %3, %4, off offset:96\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(q0[0]), "=&v"(q0[1]), "=&v"(q0[2]), "=&v"(q0[3]) : "v"(qp) : "memory");
    }
    __builtin_amdgcn_s_barrier();
    v8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = __builtin_bit_cast(v8, q0[s]);
    // Q' = Q * scale * log2(e), rounded to the operand type once per query block: the MFMA then delivers exp2 arguments
    auto prescale_q = [&](v8 (&q)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) q[s][j] = (T)((float)q[s][j] * a.scale_log2e);
    };
    prescale_q(qf);
    ASTAMP(1);
    ASTAMP(2);

    const int fr = lane & 31, fh = lane >> 5;
    const int li = lane & 15, dgrp = (lane >> 4) & 1;
    unsigned ka0[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        ka0[s] = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(sK + fr * 128 + (((2 * s + fh) ^ ((fr >> 1) & 7)) << 4));
    unsigned vb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
        vb[i] = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(sV + (4 * fh + (li >> 2)) * 128 + ((64 * i) ^ (((li >> 3) & 1) << 6)) +
                                                                                     32 * dgrp + 8 * (li & 3));
    const int nfull = CAUSAL ? 0 : a.T >> 7;                  // chunks whose 128 keys are all valid (image tower: no mask at all)
    const int tail_keys = a.T - (nfull << 7);                 // 0 .. 127 keys in the last, partial chunk
    T* obase = (T*)a.out + (int64_t)b * a.T * a.ld_out + h * 64;
#ifdef LECLIP_ATTN_ONEWAVE   // experiment (timing of ONE active wave per SIMD, same code): waves 4 .. 7 stage K / V and then idle
    constexpr int QSTEP = 4;
    if (wave >= 4) return;
#else
    constexpr int QSTEP = 8;
#endif
    for (int qb = wave; qb < nqb; qb += QSTEP) {
        const bool has_next = qb + QSTEP < nqb;
        v8 qn[4];
        if (has_next) attn_load_q<T>(a, base, qb + QSTEP, lane, qn);
        const int qi = qb * 32 + fr;
        const int qrow = qi < a.T ? qi : a.T - 1;
        float m_run = -3.0e38f, l_part = 0.f;
        f32x16 o[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
        f32x16 negm;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] = 0.f;
        auto chunk_at = [&](int key0, auto nt, auto mask, auto first, int klimit) {
            unsigned ka[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) ka[s] = ka0[s] + key0 * 128;
            stream_chunk_lazy<T, decltype(nt)::value, decltype(mask)::value, decltype(first)::value != 0>(
                ka, vb[0] + key0 * 128, vb[1] + key0 * 128, qf, o, negm, m_run, l_part, klimit - key0 - 4 * fh);
        };
        auto tail_at = [&](int key0, int keys, auto mask, auto first, int klimit) {   // a chunk with 1 .. 128 keys that may hold valid ones
            const int nt = (keys + 31) >> 5;                                         // wave-uniform
            if (nt == 1) chunk_at(key0, IntC<1>{}, mask, first, klimit);
            else if (nt == 2) chunk_at(key0, IntC<2>{}, mask, first, klimit);
            else if (nt == 3) chunk_at(key0, IntC<3>{}, mask, first, klimit);
            else chunk_at(key0, IntC<4>{}, mask, first, klimit);
        };
        if constexpr (!CAUSAL) {
            if (nfull > 0) {
                chunk_at(0, IntC<4>{}, IntC<0>{}, IntC<1>{}, a.T - 1);
                for (int ch = 1; ch < nfull; ++ch) chunk_at(ch << 7, IntC<4>{}, IntC<0>{}, IntC<0>{}, a.T - 1);
                if (tail_keys > 0) tail_at(nfull << 7, tail_keys, IntC<1>{}, IntC<0>{}, a.T - 1);
            } else {
                tail_at(0, tail_keys, IntC<1>{}, IntC<1>{}, a.T - 1);
            }
        } else {
            // causal: keys 0 .. qrow; the block's last query is min(qb * 32 + 31, T - 1) (wave-uniform bound on the chunks visited)
            const int last_q = qb * 32 + 31 < a.T - 1 ? qb * 32 + 31 : a.T - 1;
            tail_at(0, last_q + 1 < 128 ? last_q + 1 : 128, IntC<2>{}, IntC<1>{}, qrow);
            for (int key0 = 128; key0 <= last_q; key0 += 128) {
                const int keys = last_q + 1 - key0 < 128 ? last_q + 1 - key0 : 128;
                tail_at(key0, keys, IntC<2>{}, IntC<0>{}, qrow);
            }
        }
        ASTAMP(3 + 2 * (qb >> 3));
        // the next block's Q (requested a whole block ago) moves in BEFORE this block's stores: the compiler's wait for it then
        // drains nothing else (behind the stores - their number is exec-dependent to the compiler - it is a vmcnt(0) that waits for them too)
        if (has_next) {
#pragma unroll
            for (int s = 0; s < 4; ++s) qf[s] = qn[s];
            prescale_q(qf);
        }
        const float inv = 1.0f / lane32_sum(l_part);
        attn_store_block<T>(o, inv, obase + (int64_t)qi * a.ld_out, fh, qi < a.T);
        ASTAMP(4 + 2 * (qb >> 3));
    }
    ASTAMP(12);
    ASTAMP_RT(15);
}

#ifdef LECLIP_ATTN_STAMPS
extern "C" int leclip_attn_stamps_read(unsigned long long* host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_stamps), n * sizeof(unsigned long long));
}
#endif

// ---------------------------------------------------------------- fp32 validation kernel
// One workgroup per (batch, head), K and V in LDS as fp32 (rows padded to 65 floats), one query row per wave at
// a time: lanes own keys for the scores, then own output dimensions for P.V.  T <= 304 (LDS budget).
constexpr int F32_TMAX = 304;    // K and V both in LDS
constexpr int F32_TMAX_VG = 588; // K in LDS, V read from global memory (L2-resident: 64 lanes x 4 B = one 256-B row per key)

__global__ __launch_bounds__(256) void attn_f32_kernel(AttnArgs a, int v_global) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    const int TPAD = (a.T + 63) & ~63;
    float* sK = smf;                                             // [T][65]
    float* sV = smf + (size_t)a.T * 65;                          // [T][65] unless v_global
    float* sP = v_global ? sV : sV + (size_t)a.T * 65;           // [4][TPAD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / a.heads, h = blockIdx.x - b * a.heads;
    const int d_model = a.heads * 64;
    const float* base = (const float*)a.qkv + (int64_t)b * a.T * a.ld_qkv + h * 64;
    for (int i = tid; i < a.T * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        sK[r * 65 + c] = base[(int64_t)r * a.ld_qkv + d_model + c];
        if (!v_global) sV[r * 65 + c] = base[(int64_t)r * a.ld_qkv + 2 * d_model + c];
    }
    __syncthreads();
    const float scale = a.scale_log2e * 0.6931471805599453f;
    float* myP = sP + wave * TPAD;
    constexpr int NKK = (F32_TMAX_VG + 63) / 64;
    const int q_end = a.q_rows > 0 && a.q_rows < a.T ? a.q_rows : a.T;
    for (int q = wave; q < q_end; q += 4) {
        const float qd = base[(int64_t)q * a.ld_qkv + lane];   // q[d = lane]
        const int klimit = a.causal ? q : a.T - 1;
        float sloc[NKK];
        float mx = -3.0e38f;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const int key = kk * 64 + lane;
            float s = -3.0e38f;
            if (kk * 64 < a.T) {
                float dot = 0.f;
                const int kr = key < a.T ? key : a.T - 1;
                for (int d = 0; d < 64; ++d) dot = fmaf(__shfl(qd, d), sK[kr * 65 + d], dot);
                if (key <= klimit) s = dot * scale;
            }
            sloc[kk] = s;
            mx = fmaxf(mx, s);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const int key = kk * 64 + lane;
            if (kk * 64 < a.T) {
                const float p = key <= klimit ? expf(sloc[kk] - mx) : 0.f;
                if (key < TPAD) myP[key] = p;
                sum += p;
            }
        }
        sum = wave_sum(sum);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        float acc = 0.f;
        if (v_global) {
            const float* vp = base + 2 * d_model + lane;
            for (int key = 0; key <= klimit; ++key) acc = fmaf(myP[key], vp[(int64_t)key * a.ld_qkv], acc);
        } else {
            for (int key = 0; key <= klimit; ++key) acc = fmaf(myP[key], sV[key * 65 + lane], acc);
        }
        ((float*)a.out)[((int64_t)b * a.T + q) * a.ld_out + h * 64 + lane] = acc / sum;
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename T>
int launch_rows(const AttnArgs& a, int64_t B, hipStream_t s) {
    const unsigned grid = (unsigned)(B * a.heads);
#ifdef LECLIP_DIAG
    static const int force_pipe = [] { const char* e = getenv("LECLIP_ATTN_PIPE"); return e ? atoi(e) : -1; }();   // A/B timing
#else
    constexpr int force_pipe = -1;
#endif
#ifdef LECLIP_ATTN_NO_PIPE     // A/B builds: every T <= 224 call on attn_rows_kernel (one 256-thread workgroup per (batch, head), two resident per CU)
    const bool pipe_ok = false;
#else
    const bool pipe_ok = a.T > 192 && a.T <= 224;   // 7 query blocks for 8 waves
#endif
    if (pipe_ok && a.q_rows == 0 && (force_pipe == 1 || (force_pipe != 0 && grid >= 1024))) {   // enough heads to keep every CU busy
        const int n_cu = leclip_cu_count();
        static bool attr_set[LECLIP_MAX_DEVICES] = {};
        constexpr int LDSB = 4 * 7 * 32 * 128 + 8 * 4096;   // 2 x (K|V) buffers + 8 wave-private Q images
#ifdef LECLIP_ATTN_GRID_MULT    // A/B builds: more workgroups than CUs
        const unsigned cap_ = (unsigned)n_cu * LECLIP_ATTN_GRID_MULT;
        const dim3 g(grid < cap_ ? grid : cap_);
#else
        const dim3 g(grid < (unsigned)n_cu ? grid : (unsigned)n_cu);
#endif
        const int live = (a.T - 192 + 7) >> 3;   // 8-key groups of the seventh key tile that hold a valid key (T = 197: one)
#ifdef LECLIP_ATTN_NO_DEAD_GROUPS   // A/B builds: the round-4 kernel (every group of the last tile computed and masked)
        const int live_k = 4 + 0 * live;
#else
        const int live_k = live <= 1 ? 1 : live <= 2 ? 2 : 4;
#endif
        if (live_k == 1) {
            static bool attr1[LECLIP_MAX_DEVICES] = {};
            leclip_set_max_lds(attn_heads_kernel<T, 7, 1>, LDSB, attr1);
            hipLaunchKernelGGL((attn_heads_kernel<T, 7, 1>), g, dim3(512), LDSB, s, a, (int)grid);
        } else if (live_k == 2) {
            static bool attr2[LECLIP_MAX_DEVICES] = {};
            leclip_set_max_lds(attn_heads_kernel<T, 7, 2>, LDSB, attr2);
            hipLaunchKernelGGL((attn_heads_kernel<T, 7, 2>), g, dim3(512), LDSB, s, a, (int)grid);
        } else {
            leclip_set_max_lds(attn_heads_kernel<T, 7, 4>, LDSB, attr_set);
            hipLaunchKernelGGL((attn_heads_kernel<T, 7, 4>), g, dim3(512), LDSB, s, a, (int)grid);
        }
        return leclip_check_launch("attn_heads_kernel");
    }
    if (a.T <= 32) hipLaunchKernelGGL((attn_rows_kernel<T, 1>), dim3(grid), dim3(256), 0, s, a);
    else if (a.T <= 96) hipLaunchKernelGGL((attn_rows_kernel<T, 3>), dim3(grid), dim3(256), 0, s, a);
    else if (a.T <= 224) hipLaunchKernelGGL((attn_rows_kernel<T, 7>), dim3(grid), dim3(256), 0, s, a);
    else if (a.T <= STREAM_TMAX) {
        const int TP = (a.T + 127) & ~127;
        static bool attr_set[LECLIP_MAX_DEVICES] = {};
        static bool attr_set_c[LECLIP_MAX_DEVICES] = {};
        if (a.causal) {
            leclip_set_max_lds(attn_stream_kernel<T, true>, STREAM_TMAX * 256, attr_set_c);
            hipLaunchKernelGGL((attn_stream_kernel<T, true>), dim3(grid), dim3(512), TP * 256, s, a, TP);
        } else {
            leclip_set_max_lds(attn_stream_kernel<T, false>, STREAM_TMAX * 256, attr_set);
            hipLaunchKernelGGL((attn_stream_kernel<T, false>), dim3(grid), dim3(512), TP * 256, s, a, TP);
        }
        return leclip_check_launch("attn_stream_kernel");
    } else {
        leclip_set_error("attention: T=%d > %d is not supported in 16-bit modes", a.T, STREAM_TMAX);
        return LECLIP_E_UNSUPPORTED;
    }
    return leclip_check_launch("attn_rows_kernel");
}

}  // namespace

extern "C" int leclip_attention_fwd(const void* qkv, void* out, int64_t B, int T, int heads, int head_dim,
                                    int64_t ld_qkv, int64_t ld_out, leclip_mask mask, float scale,
                                    leclip_dtype dtype, void* stream) {
    return leclip_attention_prefix_fwd(qkv, out, B, T, heads, head_dim, ld_qkv, ld_out, mask, scale, 0, dtype, stream);
}

extern "C" int leclip_attention_prefix_fwd(const void* qkv, void* out, int64_t B, int T, int heads, int head_dim,
                                           int64_t ld_qkv, int64_t ld_out, leclip_mask mask, float scale, int q_rows,
                                           leclip_dtype dtype, void* stream) {
    if (q_rows < 0 || q_rows > T) { leclip_set_error("attention: q_rows %d outside [0, T=%d]", q_rows, T); return LECLIP_E_INVALID; }
    if (!qkv || !out || B <= 0 || T <= 0 || heads <= 0 || ld_qkv < 3 * heads * 64 || ld_out < heads * 64) {
        leclip_set_error("attention: null pointer or inconsistent sizes");
        return LECLIP_E_INVALID;
    }
    if (head_dim != 64) { leclip_set_error("attention: head_dim must be 64 (got %d)", head_dim); return LECLIP_E_UNSUPPORTED; }
    if (!dtype_ok(dtype) || (mask != LECLIP_MASK_NONE && mask != LECLIP_MASK_CAUSAL)) {
        leclip_set_error("attention: bad enum"); return LECLIP_E_INVALID;
    }
    if (B * heads > 0x7fffffff) { leclip_set_error("attention: grid too large"); return LECLIP_E_UNSUPPORTED; }
    AttnArgs a;
    a.qkv = qkv; a.out = out; a.T = T; a.heads = heads; a.ld_qkv = ld_qkv; a.ld_out = ld_out;
    a.scale_log2e = scale * 1.4426950408889634f; a.causal = mask == LECLIP_MASK_CAUSAL;
    a.q_rows = q_rows == T ? 0 : q_rows;
    a.reverse = leclip_walk_order() != 0;      // default (-1) and 1: the last (batch, head) pairs first (the qkv GEMM in front of it walks ascending by default)
#ifdef LECLIP_DIAG
    a.wglog = WgLog{g_leclip_wglog, g_leclip_wglog_cap, g_leclip_wglog ? ++g_leclip_wglog_seq : 0u};
#endif
    hipStream_t s = (hipStream_t)stream;
    if (dtype == LECLIP_F32) {
        if (T > F32_TMAX_VG) { leclip_set_error("attention(f32): T=%d > %d", T, F32_TMAX_VG); return LECLIP_E_UNSUPPORTED; }
        if ((uintptr_t)qkv & 3) { leclip_set_error("attention(f32): misaligned"); return LECLIP_E_INVALID; }
        const int TPAD = (T + 63) & ~63;
        const int v_global = T > F32_TMAX;
        const size_t lds = ((size_t)T * 65 * (v_global ? 1 : 2) + 4 * TPAD) * sizeof(float);
        static bool attr_set[LECLIP_MAX_DEVICES] = {};
        leclip_set_max_lds(attn_f32_kernel, 160 * 1024, attr_set);
        hipLaunchKernelGGL(attn_f32_kernel, dim3((unsigned)(B * heads)), dim3(256), lds, s, a, v_global);
        return leclip_check_launch("attn_f32_kernel");
    }
    if ((ld_qkv % 8) || (ld_out % 4) || ((uintptr_t)qkv & 15) || ((uintptr_t)out & 7)) {
        leclip_set_error("attention: qkv must be 16-byte aligned (ld %% 8 == 0), out 8-byte aligned (ld %% 4 == 0)");
        return LECLIP_E_INVALID;
    }
    return dtype == LECLIP_BF16 ? launch_rows<bf16_t>(a, B, s) : launch_rows<f16_t>(a, B, s);
}
