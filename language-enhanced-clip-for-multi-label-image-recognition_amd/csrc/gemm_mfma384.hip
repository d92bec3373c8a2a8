// Y = act(A . W^T + bias) + residual, large-M throughput kernel of round 5: 384 x 256 output tile, 8 waves, ping-pong schedule.
//
// "gemm_tn_384x256x32_pp".  Why a third tile shape: every 256 x 256 kernel measured on this chip - this build's ping-pong loop, its 4-wave / AGPR
// variant, the vendor's MT256x256x64 - ends at 0.45 - 0.53 of the MFMA peak in its K loop, and eleven schedule variants of ours moved nothing.  The
// K loop is bound by what a CU can take in from L2 (a 256 x 256 x 64 K-tile is 64 KiB of operands per 2 048 MFMA cycles; at the measured 56 % duty
// that is the ~35 B / cycle / CU of MI355X_MICROARCH.md's "gather into LDS" row), so the lever is operand bytes per FLOP ~ 1/TM + 1/TN:
// 384 x 256 moves 17 % fewer bytes per FLOP than 256 x 256 and a fifth fewer LDS fragment bytes (14 fragment reads per 48 MFMAs instead of 8 per
// 16).  tools/kloop_probe2.hip, K loop alone, M = 50 432: qkv 144 -> 135 us, out-proj 57 -> 50, c_fc 199 -> 184, c_proj 218 -> 198; the FULL-ROW
// 128 x 768 tile of the round-4 verdict moves 17 % MORE bytes per FLOP and measures 8 - 19 % slower than 256 x 256 (profiles/r05_kloop_probe2.txt).
// 384 x 256 rather than 256 x 384: N % 256 == 0 is what every model width satisfies (1024 is no multiple of 384), and a wave's 128 columns are
// two whole 64-column LayerNorm-partial slots.
//
// One 512-thread workgroup per CU; waves 4 (M) x 2 (N), each a 96 x 128 fp32 accumulator block (192 VGPRs) on v_mfma_f32_16x16x32 with
// ascending K - the instruction and K order of the 128 x 128 and 256 x 256 families, so all three produce the same bits.  Waves w and w + 4 share
// a SIMD and run the same program one barrier apart (ping-pong): one in its 24-MFMA cluster, the other in its memory cluster.
// K step = 32 (64-byte rows): an A slot (384 rows, 24 KiB) + a B slot (256 rows, 16 KiB); a ring of THREE steps (120 KiB).  A step is two phases:
//     phase 0: read the 6 A fragments + B fragments 0..3 ; stage A of step s+2 (3 pieces)      -> 24 MFMAs on columns 0..63 of the wave
//     phase 1: read B fragments 4..7                     ; stage B of step s+2 (2 pieces) ; counted wait -> 24 MFMAs on columns 64..127
// Step s+2 goes into the slot step s-1 used: its A was last read in phase (s-1, 0), its B in phase (s-1, 1) - a whole step (four barrier
// intervals) before the restage by either wave group.  RAW: the counted wait of phase (s, 1) (vmcnt(5): only step s+2's own pieces stay in flight)
// precedes that phase's barriers; step s+1 is read one phase later.  Ring slots rotate as three scalar byte offsets, so any K >= 96 with
// K % 32 == 0 runs, and the K loop runs on ACROSS output tiles (the last two steps stage the next tile's first two).
// LDS-DMA through buffer descriptors rebuilt per tile on the scalar unit (base = the tile's first row, size = its valid rows: rows past M read as
// zero), one 32-bit per-lane offset per piece, the K offset in the scalar offset.  Same 64-byte-row XOR swizzle as the 256 x 256 kernel.
//
// Epilogues: the 256 x 256 kernel's "T16" arithmetic re-cut for 16 x 128 strips (6 per wave), in TWO passes - MFMA operands swapped so that a lane
// holds 4 consecutive columns; pass 1, column block by column block: bias / fused LayerNorm / QuickGELU in the accumulator layout with the block's
// constants read from LDS once per tile, the branch value rounded to T in place (4 registers become 2) and the freed registers taking the first
// residual chunks; pass 2, strip by strip: parked through ONE wave-private LDS strip (a wave's LDS operations execute in order: write ->
// read-back -> next write need no waits), read back 16 lanes per row, 16 bytes per lane; 16-bit residual added after the read-back (the
// reference's two roundings, clip/model.py:225-228), LayerNorm partials from the stored values with the 256 x 256 kernel's additions in its
// order.  Column constants and the tile's (mean, rstd) rows travel by LDS-DMA into a constants region ahead of the last two steps.
// Flavours: residual (+ partials) [out-proj, c_proj: clip/model.py:226-227], fused LayerNorm (+ QuickGELU) [qkv, c_fc: :212-218, :221-223], plain bias.
// With EpiParams::stats_merged (leclip_gemm_res_stats_fwd) the LAST of a 384-row block's N / 256 workgroups to finish merges the block's partials
// into (mean, rstd) inside the launch - what the next LayerNorm's statistics pass (clip/model.py:193-199) would compute.
#include "leclip_common.h"
#include <stdlib.h>

namespace {

constexpr int TM = 384, TN = 256, KS = 32;
constexpr int A_SLOT = TM * 64, B_SLOT = TN * 64, PAIR = A_SLOT + B_SLOT, RING = 3 * PAIR;   // 24 + 16 = 40 KiB per step, 120 KiB
constexpr int PITCH = 264, STRIP = 16 * PITCH;                                                  // 16 x 128 strip of 16-bit values, 8 bytes of row padding
constexpr int STRIPS_OFF = RING, CST_OFF = RING + 8 * STRIP;                                    // constants: bias[256] | colsum[256] | (mean, rstd)[384]
constexpr int LDS_BYTES = CST_OFF + 2048 + 3072;                                                // 161 792 of 163 840
constexpr int PIECES = 5;                                                                       // LDS-DMA pieces per wave and step (3 A + 2 B)

typedef __attribute__((ext_vector_type(4))) float acc4;
typedef __attribute__((ext_vector_type(2))) int i32x2;

__device__ __forceinline__ acc4 mfma16(bf16x8 a, bf16x8 b, acc4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ acc4 mfma16(f16x8 a, f16x8 b, acc4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

struct Gemm384Args {
    const void* A;
    const void* W;
    int64_t M;
    int N, K;
    int64_t lda, ldw;
    EpiParams epi;
    int tiles_n, tiles_total;
    int reverse;   // walk the tiles from the last to the first (leclip_set_walk_order)
#ifdef LECLIP_DIAG
    WgLog wglog;   // diagnostic library only: per-workgroup begin / end log (profiles/two_part_timeline.py)
#endif
};

__device__ __forceinline__ int xcd_remap384(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

#define PIN() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ char* uniform_ptr384(const char* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (char*)(((unsigned long long)hi << 32) | lo);
}

// The K loop's state: ring offsets, staging descriptors, accumulators.
template <typename T>
struct KL {
    typedef typename VecOf<T>::v8 v8;
    char* smem;
    int a_rd, b_rd;            // per-lane LDS byte offsets of the fragment reads inside a step's slot pair
    int wave1k;                // wave * 1024: this wave's first piece inside a slot
    unsigned va[3], vw[2];     // per-lane source byte offsets of the wave's pieces (row * leading dimension + swizzled 16-byte chunk)
    int cur, n1, n2;           // ring: byte offsets of the slot pairs holding step s, s+1 and the one being staged (s+2)
    __amdgpu_buffer_rsrc_t ra, rw;
    acc4 acc[6][8];
    v8 af[6], bfr[4];

    __device__ __forceinline__ void stage_a(int slot, int kbyte) {
#pragma unroll
        for (int u = 0; u < 3; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(smem + slot + wave1k + u * 8192), 16, (int)va[u], kbyte, 0, 0);
    }
    __device__ __forceinline__ void stage_b(int slot, int kbyte) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(smem + slot + A_SLOT + wave1k + u * 8192), 16, (int)vw[u], kbyte, 0, 0);
    }
    __device__ __forceinline__ void read_a(int slot) {
        const char* p = smem + slot + a_rd;
#pragma unroll
        for (int i = 0; i < 6; ++i) af[i] = *(const v8*)(p + i * 1024);
    }
    __device__ __forceinline__ void read_b(int slot, int h) {
        const char* p = smem + slot + b_rd + h * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = *(const v8*)(p + j * 1024);
    }
    // operands swapped (W fragment first): block (i, j) of lane l holds C[16 i + (l & 15)][16 j + 4 (l >> 4) + r]
    __device__ __forceinline__ void compute(int h) {
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PIN();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][4 * h + j] = mfma16(bfr[j], af[i], acc[i][4 * h + j]);
        __builtin_amdgcn_s_setprio(0);
        PIN();
        __builtin_amdgcn_s_barrier();
        PIN();
    }
    __device__ __forceinline__ void rotate() {
        const int t = cur;
        cur = n1;
        n1 = n2;
        n2 = t;
    }
    // One step, ONE copy of the code for every position in a tile (peeled copies of the last two steps made the register allocator time-share
    // accumulator registers there: scratch traffic, and a compiler vmcnt(0) in front of every reload).  st: stage (step s+2 of this tile, or the
    // next tile's step 0 / 1 from the last two steps when there is a next tile).  wsel picks the counted wait: 0 = vmcnt(5) (only the pieces just
    // staged stay in flight), 1 = vmcnt(5 + X) (first step of a tile: the previous epilogue's X operations are younger than the pieces waited
    // for), 2 = vmcnt(0) (second to last step of the last tile: its last operands and the epilogue constants), 3 = none.
    template <int X>
    __device__ __forceinline__ void step(int kbyte, bool st, int wsel) {
        read_a(cur);
        read_b(cur, 0);
        if (st) stage_a(n2, kbyte);
        PIN();
        compute(0);
        read_b(cur, 1);
        if (st) stage_b(n2, kbyte);
        if (wsel == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        else if (wsel == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES + X) : "memory");
        else if (wsel == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PIN();
        compute(1);
        rotate();
    }
};

// PF: 0 no residual / LayerNorm, 1 16-bit residual, 2 fused LayerNorm.  CFG: bit 0 QuickGELU, bit 1 emit LayerNorm partials of the output.
template <typename T, int PF, int CFG>
__global__ __launch_bounds__(512, 2) void gemm_tn_384x256x32_pp(Gemm384Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ACT = CFG & 1, STATS = (CFG >> 1) & 1;
    typedef typename VecOf<T>::v4 v4t;
    typedef typename VecOf<T>::v8 v8t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wm = wave >> 1, wn = wave & 1;
    const EpiParams& e = g.epi;
    const int ns = g.K / KS;   // >= 3 (host)

#ifdef LECLIP_DIAG
    unsigned long long wl_t0 = 0, wl_c0 = 0;
    if (g.wglog.buf && tid == 0) { wl_t0 = __builtin_amdgcn_s_memrealtime(); wl_c0 = __builtin_amdgcn_s_memtime(); }
#endif
    KL<T> p;
    p.smem = smem;
    p.wave1k = wave * 1024;
    // per-lane constants of the K loop, recomputed at every tile top from a laundered lane id so that they do not live through the epilogue
    auto lane_constants = [&]() {
        int l = lane;
        asm volatile("" : "+v"(l));
        const int fr = l & 15, fc = l >> 4;
        const int frd = fr * 64 + ((fc ^ ((4 - ((fr >> 2) & 3)) & 3)) << 4);
        p.a_rd = wm * (96 * 64) + frd;
        p.b_rd = A_SLOT + wn * (128 * 64) + frd;
        // piece rows: lane l writes row 16 * piece + (l >> 2), physical chunk l & 3, which must hold logical chunk (l & 3) ^ f((row >> 2) & 3)
        const unsigned dma_c = ((l & 3) ^ ((4 - ((l >> 4) & 3)) & 3)) * 16, dma_r = l >> 2;
        const unsigned ldab = (unsigned)g.lda * 2, ldwb = (unsigned)g.ldw * 2;
#pragma unroll
        for (int u = 0; u < 3; ++u) p.va[u] = ((wave + 8 * u) * 16 + dma_r) * ldab + dma_c;
#pragma unroll
        for (int u = 0; u < 2; ++u) p.vw[u] = ((wave + 8 * u) * 16 + dma_r) * ldwb + dma_c;
    };
    lane_constants();
    p.cur = 0;
    p.n1 = PAIR;
    p.n2 = 2 * PAIR;

    auto tile_origin = [&](int v, int64_t& m0, int& n0) {
        if (g.reverse) v = g.tiles_total - 1 - v;
        const int tile = xcd_remap384(v, g.tiles_total);
        int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
#ifdef LECLIP_G384_NGROUP   // A/B builds: N-tiles in groups of NGROUP, (group, row block, tile in group) order: an XCD's chunk stays on one group's W panel
        if (g.tiles_n > LECLIP_G384_NGROUP && g.tiles_n % LECLIP_G384_NGROUP == 0) {
            const int per_group = (g.tiles_total / g.tiles_n) * LECLIP_G384_NGROUP;
            const int grp_ = tile / per_group, rem = tile - grp_ * per_group;
            tm = rem / LECLIP_G384_NGROUP;
            tn = grp_ * LECLIP_G384_NGROUP + (rem - tm * LECLIP_G384_NGROUP);
        }
#endif
        m0 = (int64_t)tm * TM;
        n0 = tn * TN;
    };
    auto set_sources = [&](int64_t m0, int n0) {
        const int64_t left = g.M - m0;
        const int rows = left >= TM ? TM : (int)left;
        p.ra = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr384((const char*)g.A + m0 * g.lda * 2), 0,
                                                 __builtin_amdgcn_readfirstlane((int)((unsigned)rows * (unsigned)g.lda * 2u)), 0x00020000);
        p.rw = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr384((const char*)g.W + (int64_t)n0 * g.ldw * 2), 0,
                                                 __builtin_amdgcn_readfirstlane((int)((unsigned)TN * (unsigned)g.ldw * 2u)), 0x00020000);
    };

    // vector-memory operations one epilogue issues per wave: 6 strips x (4 stores [+ 1 partials store] [+ 4 residual loads])
    constexpr int EPI_OPS = 6 * (4 + STATS + (PF == 1 ? 4 : 0));
    static_assert(PIECES + EPI_OPS <= 63, "vmcnt is a 6-bit counter");

    int v = blockIdx.x;
    int64_t m0;
    int n0;
    tile_origin(v, m0, n0);
    set_sources(m0, n0);
    p.stage_a(p.cur, 0);
    p.stage_b(p.cur, 0);
    p.stage_a(p.n1, 2 * KS);
    p.stage_b(p.n1, 2 * KS);
    bool first = true;
    while (true) {
        if (!first) lane_constants();
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) p.acc[i][j] = acc4{0.f, 0.f, 0.f, 0.f};
        // step 0 of this tile has landed: behind it in the queue are step 1's pieces and - after a first tile - the previous epilogue's operations
        if (first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES + EPI_OPS) : "memory");
        PIN();
        __builtin_amdgcn_s_barrier();
        PIN();
        if (grp == 1) __builtin_amdgcn_s_barrier();   // stagger: waves 4..7 run one barrier behind
        PIN();

        const int vn = v + gridDim.x;
        const bool more = vn < g.tiles_total;
        int64_t m0n = 0;
        int n0n = 0;
        if (more) tile_origin(vn, m0n, n0n);
        for (int s = 0; s < ns; ++s) {
            const bool last2 = s + 2 >= ns;
            if (s + 2 == ns) {
                // The tile's constants by LDS-DMA, ahead of the last two steps (older than the next tile's pieces, landed at this step's wait):
                // column constants 4 bytes per lane (bias: waves 0..3, LayerNorm column sums: 4..7), the rows' (mean, rstd) as three 1 KiB
                // pieces (waves 0..2).
                int t_ = tid;
                asm volatile("" : "+v"(t_));
                float* cst = (float*)(smem + CST_OFF);
                const float* cp = wave < 4 ? e.bias : (PF == 2 ? e.ln_colsum : nullptr);   // (wave-uniform)
                if (cp) __builtin_amdgcn_global_load_lds((const void*)(cp + n0 + (t_ & 255)), LDS_PTR(cst + wave * 64), 4, 0, 0);
                else if (wave < 4 || PF == 2) cst[t_] = 0.f;
                if constexpr (PF == 2) {
                    if (wave < 3) {
                        const int64_t left = g.M - m0;
                        const int rows = left >= TM ? TM : (int)left;
                        const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr384((const char*)(e.ln_stats + 2 * m0)), 0,
                                                                                            __builtin_amdgcn_readfirstlane(rows * 8), 0x00020000);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rl, LDS_PTR(smem + CST_OFF + 2048 + wave * 1024), 16, wave * 1024 + (t_ & 63) * 16, 0, 0, 0);
                    }
                }
                if (more) set_sources(m0n, n0n);
            }
            const bool st = !last2 || more;
            const int kbyte = (last2 ? s + 2 - ns : s + 2) * (2 * KS);
            const int wsel = (s == 0 && !first) ? 1 : st ? 0 : (s + 2 == ns ? 2 : 3);
            p.template step<EPI_OPS>(kbyte, st, wsel);
        }

        // ---- epilogue, two passes.
        // Pass 1, column block by column block (j outer, the six strips inner): the block's constants are read from LDS ONCE per tile, the branch
        // value (bias / fused LayerNorm / QuickGELU) is computed in the accumulator layout and rounded to T in place - 4 fp32 registers become 2 -
        // and the registers that frees take the residual chunks of strips 0..3, two loads per block: 16 are in flight when pass 1 ends, their
        // latency under its arithmetic (the strip-major first version read the constants 6 times and waited for the first chunks on the spot);
        // strips 4 and 5 follow from pass 2 as it frees registers.
        // Pass 2, strip by strip: park the 16 x 128 strip (8 ds_write_b64), read it back 16 lanes per row, add the residual, store, partials.
        const int64_t em0 = m0;
        const int en0 = n0;
        int lane_e = lane;   // laundered: what derives from it is recomputed per tile instead of living across the K loop
        asm volatile("" : "+v"(lane_e));
        const int64_t row0 = em0 + wm * 96;
        const int64_t left = g.M - row0;
        const int rows_ok = left >= 96 ? 96 : (left > 0 ? (int)left : 0);
        // residual chunks: inline-asm buffer loads with counted waits (tests/isa_audit.py checks that nothing touches a destination between a
        // load and the wait that names it); chunk 4 q + pp of this lane = row 16 q + 4 pp + (lane >> 4), columns 8 (lane & 15) ..
        i32x4 rpre[PF == 1 ? 24 : 1];
        const int ldrb = PF == 1 ? __builtin_amdgcn_readfirstlane((int)e.ldr * 2) : 0;
        i32x4 rdesc = {0, 0, 0, 0};
        if constexpr (PF == 1) {
            const unsigned long long rb = (unsigned long long)uniform_ptr384((const char*)e.res + (row0 * e.ldr + en0 + wn * 128) * 2);
            rdesc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)rb);
            rdesc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(rb >> 32) & 0xffff);
            rdesc[2] = __builtin_amdgcn_readfirstlane(rows_ok ? (rows_ok - 1) * ldrb + 256 : 0);
            rdesc[3] = 0x00020000;
        }
        const int rvoff = PF == 1 ? (lane_e >> 4) * ldrb + (lane_e & 15) * 16 : 0;
        auto load_res = [&](int first, int n) {   // chunks first .. first + n - 1
#pragma unroll
            for (int c = first; c < first + n; ++c) {
                const int off = rvoff + ((c >> 2) * 16 + (c & 3) * 4) * ldrb;
                if (c == first) asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(rpre[PF == 1 ? c : 0]) : "v"(off), "s"(rdesc) : "memory");
                else asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(rpre[PF == 1 ? c : 0]) : "v"(off), "s"(rdesc) : "memory");
            }
        };
        PIN();
        if (grp == 0) __builtin_amdgcn_s_barrier();   // re-align the two groups (equal barrier counts)
        PIN();

        v4t wq[6][8];   // the tile's branch values, rounded to T: lane (m = l & 15, g = l >> 4) holds row 16 q + m, columns 16 j + 4 g .. + 3
        {
            const float* cst = (const float*)(smem + CST_OFF);
            const int m = lane_e & 15, gq = lane_e >> 4;
            const float* cb = cst + wn * 128 + 4 * gq;
            f32x2 mr[PF == 2 ? 6 : 1];
            if constexpr (PF == 2) {
#pragma unroll
                for (int q = 0; q < 6; ++q) mr[q] = *(const f32x2*)(cst + 512 + 2 * (wm * 96 + q * 16 + m));
            }
            f32x4 bn = *(const f32x4*)cb, sn = {0.f, 0.f, 0.f, 0.f};
            if constexpr (PF == 2) sn = *(const f32x4*)(cb + 256);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 b4 = bn, s4 = sn;
                if (j + 1 < 8) {
                    bn = *(const f32x4*)(cb + 16 * (j + 1));
                    if constexpr (PF == 2) sn = *(const f32x4*)(cb + 256 + 16 * (j + 1));
                }
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    v4t w;
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        // the 256 x 256 kernel's arithmetic, two columns at a time as packed fp32 (same IEEE operations)
                        typedef float f2 __attribute__((ext_vector_type(2)));
                        f2 v = {p.acc[q][j][r], p.acc[q][j][r + 1]};
                        const f2 bb = {b4[r], b4[r + 1]};
                        if constexpr (PF == 2) {
                            const f2 ss = {s4[r], s4[r + 1]};
                            v = __builtin_elementwise_fma((f2)(mr[q][1]), __builtin_elementwise_fma((f2)(-mr[q][0]), ss, v), bb);
                        } else v += bb;
                        if constexpr (ACT == 1) {
                            const f2 t = v * (f2)(-2.4554669595930157f);
                            f2 d = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                            d += (f2)(1.0f);
                            const f2 rc = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
                            v *= rc;
                        }
                        w[r] = (T)v.x;
                        w[r + 1] = (T)v.y;
                    }
                    wq[q][j] = w;
                }
                PIN();
                if constexpr (PF == 1) load_res(2 * j, 2);   // (block j's 24 accumulator registers became 12: room for two chunks and pass 2's temporaries)
                PIN();
            }
        }

        {
            char* st = smem + STRIPS_OFF + wave * STRIP;
            const int ldb = __builtin_amdgcn_readfirstlane((int)e.ldy * 2);
            const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
                uniform_ptr384((const char*)e.out + (row0 * e.ldy + en0 + wn * 128) * 2), 0,
                __builtin_amdgcn_readfirstlane(rows_ok ? (rows_ok - 1) * ldb + 256 : 0), 0x00020000);
            __amdgpu_buffer_rsrc_t srsrc;
            if constexpr (STATS == 1)   // partials, slot-major [slot][row][2]: the wave's two slots, rows from row0; lanes address (slot, row) themselves
                srsrc = __builtin_amdgcn_make_buffer_rsrc(
                    uniform_ptr384((const char*)e.stats_out + ((int64_t)((en0 + wn * 128) >> 6) * e.stats_rows + row0) * 8), 0,
                    __builtin_amdgcn_readfirstlane((int)((e.stats_rows + rows_ok) * 8)), 0x00020000);
            int lane_q = lane_e;
            asm volatile("" : "+v"(lane_q));
            char* sq = st + (lane_q & 15) * PITCH + (lane_q >> 4) * 8;
            auto park = [&](int q) {
#pragma unroll
                for (int j = 0; j < 8; ++j) *(v4t*)(sq + j * 32) = wq[q][j];
            };
            park(0);
            const int crow = lane_q >> 4, c16 = (lane_q & 15) * 16;
            const int voff = crow * ldb + c16;
            const int cc = lane_q & 7;
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                v8t o[4];
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) {
                    const char* sp = st + (4 * pp + crow) * PITCH + c16;
                    const v4t r0 = *(const v4t*)sp, r1 = *(const v4t*)(sp + 8);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { o[pp][c] = r0[c]; o[pp][4 + c] = r1[c]; }
                }
                PIN();
                if (q + 1 < 6) park(q + 1);   // (a wave's LDS operations execute in order: behind the read-back, no wait needed)
                PIN();
                if constexpr (PF == 1) {
                    // queue: L0-15 (pass 1) | S0 | L16-19 . S1 | L20-23 . S2 | S3 | S4 | S5     (L = residual chunks, S = a strip's 4 + STATS stores)
                    if (q == 1 || q == 2) load_res(12 + 4 * q, 4);
                    constexpr int SS = 4 + STATS;
#define RES_WAIT(N)                                                                                                                                   \
    asm volatile("s_waitcnt vmcnt(%4)"                                                                                                                \
                 : "+v"(rpre[PF == 1 ? 4 * q : 0]), "+v"(rpre[PF == 1 ? 4 * q + 1 : 0]), "+v"(rpre[PF == 1 ? 4 * q + 2 : 0]), "+v"(rpre[PF == 1 ? 4 * q + 3 : 0]) \
                 : "n"(N) : "memory")
                    if (q == 0) RES_WAIT(12);
                    else if (q == 1) RES_WAIT(12 + SS);
                    else if (q == 2) RES_WAIT(12 + 2 * SS);
                    else if (q == 3) RES_WAIT(8 + 3 * SS);
                    else if (q == 4) RES_WAIT(4 + 3 * SS);
                    else RES_WAIT(3 * SS);
#undef RES_WAIT
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp) {
                        const v8t rr = __builtin_bit_cast(v8t, rpre[PF == 1 ? 4 * q + pp : 0]);
#pragma unroll
                        for (int c = 0; c < 8; ++c) o[pp][c] = (T)((float)o[pp][c] + (float)rr[c]);
                    }
                }
#pragma unroll
                for (int pp = 0; pp < 4; ++pp)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, o[pp]), orsrc, voff + (q * 16 + pp * 4) * ldb, 0, 0);
#ifdef LECLIP_G384_ABL_NOSTATS   // timing-only A/B build: no partials (out-proj 75.2 -> 72.8 us, c_proj 229.8 -> 227.2: profiles/r05_g384_notes.txt)
                if constexpr (false) {
#else
                if constexpr (STATS == 1) {
#endif
                    // LayerNorm partials of the stored values: the 8 lanes of a (row, 64-column slot) hold its pair; lane cc < 4 of each group stores
                    // the pair of pass cc's row - one store per strip: two 128-byte runs (16 rows x 8 bytes) in the wave's two slots
                    f32x2 sel = row_block_stats(o[0]);
#pragma unroll
                    for (int pp = 1; pp < 4; ++pp) {   // (one pass at a time: four interleaved cost 32 temporaries the residual flavours do not have)
                        PIN();
                        const f32x2 sp = row_block_stats(o[pp]);
                        sel = cc == pp ? sp : sel;
                    }
                    const int r = q * 16 + cc * 4 + crow;
                    const int svoff = (cc < 4 && r < rows_ok) ? (int)((((lane_q >> 3) & 1) * e.stats_rows + r) * 8) : 0x7ff00000;
                    // (write-through, sc1: the row block's last-arriving workgroup may read these pairs in this launch - see the merge below)
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, sel), srsrc, svoff, 0, 16);
                }
            }
        }
        if constexpr (STATS == 1) {
            if (e.stats_merged) {   // (uniform)
                // In-producer LayerNorm merge.  Hand-off by MI355X_MICROARCH.md's measured form: every byte stored sc1 and drained by its storing
                // wave (vmcnt(0)), a workgroup barrier, ONE lane's returning agent-scope add on the row block's counter; the workgroup whose add
                // comes last (told by the returned value - nobody waits or spins) reads every pair with sc1 loads.  Both barriers are executed
                // by all eight waves (the two wave groups are aligned here).
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                unsigned* flag = (unsigned*)(smem + CST_OFF + 2048);      // (the (mean, rstd) region: unused by the residual flavours)
                const int tmb = (int)(em0 / TM);
                if (wave == 0) {
                    unsigned old = 0;
                    if (lane == 0) old = __hip_atomic_fetch_add(e.stats_tickets + tmb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    old = __builtin_amdgcn_readfirstlane(old);
                    const unsigned last = old == (unsigned)(g.tiles_n - 1) ? 1u : 0u;
                    if (lane == 0) {
                        *flag = last;
                        if (last) __hip_atomic_store(e.stats_tickets + tmb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                if (*(volatile __attribute__((address_space(3))) unsigned*)LDS_PTR(flag)) {   // (an LDS read, not a flat one)
                    // 384 rows, 48 per wave, one per lane: the row's pairs slot by slot (consecutive lanes read consecutive pairs of a slot)
                    int lm = lane_e;
                    asm volatile("" : "+v"(lm));
                    const int64_t leftb = g.M - em0;
                    const int rows_b = leftb >= TM ? TM : (int)leftb;
                    const int row = wave * 48 + lm;
                    const bool act_row = lm < 48 && row < rows_b;
                    const int slots = e.stats_slots;
                    const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(
                        uniform_ptr384((const char*)e.stats_out), 0, __builtin_amdgcn_readfirstlane((int)((unsigned)slots * (unsigned)e.stats_rows * 8u)), 0x00020000);
                    const int pv = act_row ? (int)((em0 + row) * 8) : 0x7ff00000;
                    const int sstep = __builtin_amdgcn_readfirstlane((int)(e.stats_rows * 8));
                    f32x4 v[LN_MERGE_MAXV];
#pragma unroll
                    for (int i = 0; i < LN_MERGE_MAXV; ++i) {
                        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (2 * i < slots) {
                            const i32x2 a = __builtin_amdgcn_raw_buffer_load_b64(prsrc, pv, (2 * i) * sstep, 16);
                            const i32x2 b = __builtin_amdgcn_raw_buffer_load_b64(prsrc, pv, (2 * i + 1) * sstep, 16);
                            v[i][0] = __int_as_float(a[0]); v[i][1] = __int_as_float(a[1]); v[i][2] = __int_as_float(b[0]); v[i][3] = __int_as_float(b[1]);
                        }
                    }
                    const f32x2 mr = ln_merge_partials(v, slots, g.N, e.stats_eps);
                    if (act_row) *(f32x2*)(e.stats_merged + 2 * (em0 + row)) = mr;
                }
            }
        }
        if (!more) break;
        v = vn;
        m0 = m0n;
        n0 = n0n;
        first = false;
    }
#ifdef LECLIP_DIAG
    if (g.wglog.buf && tid == 0) wglog_end(g.wglog, 0x300u + (unsigned)(PF * 16 + (CFG & 15)), wl_t0, wl_c0);
#endif
}

template <typename T, int PF, int CFG>
int launch384_pf(const Gemm384Args& a, int grid, hipStream_t s) {
    static bool attr_set[LECLIP_MAX_DEVICES] = {};
    leclip_set_max_lds((gemm_tn_384x256x32_pp<T, PF, CFG>), LDS_BYTES, attr_set);
    hipLaunchKernelGGL((gemm_tn_384x256x32_pp<T, PF, CFG>), dim3(grid), dim3(512), LDS_BYTES, s, a);
    return leclip_check_launch("gemm_tn_384x256x32_pp");
}

template <typename T>
int launch384(const Gemm384Args& a, hipStream_t s) {
    const int n_cu = leclip_cu_count();
#ifdef LECLIP_G384_GRID_MULT   // A/B builds: more workgroups than CUs (0: one per tile) - the dispatcher then backfills freed CUs from either stream part's launch
    const int cap = LECLIP_G384_GRID_MULT > 0 ? n_cu * LECLIP_G384_GRID_MULT : a.tiles_total;
    const int grid = a.tiles_total < cap ? a.tiles_total : cap;
#else
    const int grid = a.tiles_total < n_cu ? a.tiles_total : n_cu;
#endif
    const EpiParams& e = a.epi;
    const bool gelu = e.act == LECLIP_ACT_QUICKGELU, stats = e.stats_out != nullptr, ln = e.ln_stats != nullptr;
    if (e.res) return stats ? launch384_pf<T, 1, 2>(a, grid, s) : launch384_pf<T, 1, 0>(a, grid, s);
    if (ln) return gelu ? launch384_pf<T, 2, 1>(a, grid, s) : launch384_pf<T, 2, 0>(a, grid, s);
    return gelu ? launch384_pf<T, 0, 1>(a, grid, s) : launch384_pf<T, 0, 0>(a, grid, s);
}

}  // namespace

// Calls this kernel takes: N % 256 == 0, K % 32 == 0, K >= 96, output (and residual) in the operand dtype, one of the hot epilogue
// combinations, and enough tiles to fill the chip.  All three families produce the same bits, so the choice is a rate decision only;
// leclip_set_gemm_family() overrides it (tests compare the families with it).
bool leclip_gemm384_shape(int64_t M, int N, int K) {
    if (N % TN != 0 || K % KS != 0 || K < 3 * KS) return false;
    const int fam = leclip_gemm_family();
    if (fam >= 0) return fam == 384;
    return ((M + TM - 1) / TM) * (N / TN) >= 160;
}

// In-producer LayerNorm merge: the partials' slots must fit ln_merge_partials (an even count <= 16) and a block's pair offsets 32 bits.
bool leclip_gemm384_merges(int64_t M, int N) { return (N / 64) % 2 == 0 && N / 64 <= 2 * LN_MERGE_MAXV && (int64_t)(N / 64) * M * 8 < 0x7ff00000LL; }

bool leclip_gemm384_eligible(int64_t M, int N, int K, int64_t lda, int64_t ldw, const EpiParams& e, int ab_dtype) {
    if (!leclip_gemm384_shape(M, N, K)) return false;
    if (lda >= (1 << 22) || ldw >= (1 << 22) || e.ldy >= (1 << 22) || (e.res && e.ldr >= (1 << 22))) return false;
    if (M * lda * 2 > 0x7fffffffLL * 2) return false;
    const bool ln = e.ln_stats != nullptr, gelu = e.act == LECLIP_ACT_QUICKGELU, stats = e.stats_out != nullptr;
    if (e.out_dt != ab_dtype || e.rowmap_P || (e.res && e.res_dt != ab_dtype)) return false;
    if (e.res && (ln || gelu)) return false;
    if (!e.res && stats) return false;
#ifdef LECLIP_G384_FLAVOURS   // A/B builds (make variant DEFS=-DLECLIP_G384_FLAVOURS=n): bit 0 residual, bit 1 fused LayerNorm, bit 2 plain
    if (leclip_gemm_family() != 384 && !((LECLIP_G384_FLAVOURS) & (e.res ? 1 : ln ? 2 : 4))) return false;
#endif
    if (ln && (((uintptr_t)e.ln_stats & 15) || !e.ln_colsum)) return false;
    return true;
}

int leclip_gemm384_launch(const void* A, const void* W, int64_t M, int N, int K, int64_t lda, int64_t ldw, const EpiParams& epi, int ab_dtype,
                          hipStream_t s) {
    Gemm384Args a;
    a.A = A; a.W = W; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.epi = epi;
    const int64_t tiles_m = (M + TM - 1) / TM;
    a.tiles_n = N / TN;
    if (tiles_m * a.tiles_n > 0x7fffffff) { leclip_set_error("gemm: too many tiles"); return LECLIP_E_UNSUPPORTED; }
    a.tiles_total = (int)(tiles_m * a.tiles_n);
    a.reverse = leclip_walk_order() == 1;
#ifdef LECLIP_DIAG
    a.wglog = WgLog{g_leclip_wglog, g_leclip_wglog_cap, g_leclip_wglog ? ++g_leclip_wglog_seq : 0u};
#endif
    return ab_dtype == LECLIP_BF16 ? launch384<bf16_t>(a, s) : launch384<f16_t>(a, s);
}
