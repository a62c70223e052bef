// K11 deterministic form: the gather backward (autograd of model/GaussNet.py:63-83) as a gather-REDUCE
// over an inverted index instead of a scatter with float atomics.
//
// The 8-NN index map of a view is static (it is built once by create_index_and_dist and reused by every
// attack epoch), so its inverse - for each row j of the perturbation table, the list of (pixel, k) that
// gather from it - is built once (radix sort of (destination, contribution id) pairs, rocPRIM via hipCUB)
// and reused. The backward is then
//     pass 1 (per pixel, streaming):  g[p] = dL/dx[p]  (chain through alpha / epsilon clip / where / clip)
//     pass 2 (per destination row):   grad_s[j] = sum_{c in row j} w_c * g[pixel_c]     in a FIXED order
// => bitwise reproducible, no atomics (the MI355X float-atomic rate for 16-byte scattered segments is
// ~0.08 TB/s; the gathers here are served by L2 / Infinity Cache).
#include "common.h"

#include <hipcub/hipcub.hpp>

namespace nerfail {

__global__ __launch_bounds__(256) void csr_keys_kernel(const float* __restrict__ wi, long Ns, long B, long P,
                                                       unsigned* __restrict__ keys, int* __restrict__ vals) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;     // contribution id = (b*P + p)*8 + k
    if (g >= B * P * 8) return;
    const long bp = g >> 3;
    const int k = (int)(g & 7);
    const long b = bp / P, p = bp - b * P;
    long j = (long)wi[((b * 2 + 1) * P + p) * 8 + k];
    j = j < 0 ? 0 : (j >= Ns ? Ns - 1 : j);
    // A contribution with weight 0 adds w * g = 0 to its row: it is left out of the index (key Ns sorts it behind the last
    // row, row_ptr[Ns] ends before it). On real maps that is most of them: a background pixel's 3-D point (near plane,
    // NC:418-423) is far from every point of the set, all 8 Gaussian weights underflow to 0 (GN:181), and its 8
    // "neighbours" are the same few boundary points for hundreds of thousands of pixels - rows that long kept one wave
    // busy for milliseconds after the rest of the grid had finished. (The atomic form skips w == 0 as well.)
    const float w = wi[((b * 2 + 0) * P + p) * 8 + k];
    keys[g] = (w == 0.f) ? (unsigned)Ns : (unsigned)j;   // one list per destination row over the WHOLE batch (ids ascending inside: b, p, k)
    vals[g] = (int)g;
}

// row_ptr[r] = first sorted position whose key >= r (r in [0, Ns]); weights gathered into sorted order
__global__ __launch_bounds__(256) void csr_rows_kernel(const unsigned* __restrict__ keys_sorted, long n, long rows,
                                                       int* __restrict__ row_ptr) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > rows) return;
    long lo = 0, hi = n;
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if ((long)keys_sorted[mid] < r) lo = mid + 1; else hi = mid;
    }
    row_ptr[r] = (int)lo;
}

__global__ __launch_bounds__(256) void csr_weights_kernel(const float* __restrict__ wi, const int* __restrict__ contrib,
                                                          long n, long P, float* __restrict__ w_sorted) {
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const long g = contrib[c];
    const long bp = g >> 3;
    const long b = bp / P, p = bp - b * P;
    w_sorted[c] = wi[((b * 2 + 0) * P + p) * 8 + (g & 7)];
}

// pass 1: effective dL/dx per pixel (same chain as gauss_bwd_kernel in gauss.hip)
__global__ __launch_bounds__(256) void gauss_pixel_grad_kernel(const float4* __restrict__ ori, const float4* __restrict__ x_saved,
                                                               const float4* __restrict__ grad_x,
                                                               const float4* __restrict__ grad_x_rgba, long n, float epsilon,
                                                               float4* __restrict__ g_out) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const float4 o = ori[g];
    float4 gx = (grad_x != nullptr) ? grad_x[g] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (grad_x_rgba != nullptr && o.w > 0.f) {                 // (transparent pixels - most of a view - read nothing else)
        const float4 x = x_saved[g];
        const float4 gr = grad_x_rgba[g];
        const float alpha = x.w / 255.0f;
        const float xc[3] = {x.x, x.y, x.z}, oc[3] = {o.x, o.y, o.z}, grc[3] = {gr.x, gr.y, gr.z};
        float gxc[3] = {0.f, 0.f, 0.f};
        float ga = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float d = xc[c] * alpha;
            bool pass = true;
            if (epsilon >= 0.f) {
                pass = (d >= -epsilon) && (d <= epsilon);
                d = fminf(fmaxf(d, -epsilon), epsilon);
            }
            const float pre = oc[c] + d;
            pass = pass && (pre >= 0.f) && (pre <= 255.f);
            const float gd = pass ? grc[c] : 0.f;
            gxc[c] = gd * alpha;
            ga += gd * xc[c];
        }
        gx.x += gxc[0]; gx.y += gxc[1]; gx.z += gxc[2];
        gx.w += ga / 255.0f;
    }
    g_out[g] = gx;
}

// pass 1, rgb-gradient-only form (the NeRFail-S step: AS:357-392 never reads the alpha channel's gradient): from what the
// forward left behind - alpha and the 3-bit pass mask, 5 bytes per pixel - instead of x and ori (32 bytes). Pixels whose
// mask is 0 (background, saturated) read nothing else. The rgb values equal gauss_pixel_grad_kernel's bit for bit.
// (Round 4 measured the alternative the round-3 verdict asked for - forming this at the gather inside the reduce, no g_pix
// round trip: three gathers per entry instead of one, reduce 98 -> 143 us for the 25 us this pass takes. Kept as a pass.)
__global__ __launch_bounds__(256) void gauss_pixel_grad_rgb_kernel(const float* __restrict__ aux_alpha,
                                                                   const unsigned char* __restrict__ aux_mask,
                                                                   const float4* __restrict__ grad_x_rgba, long n,
                                                                   float4* __restrict__ g_out) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const unsigned m = aux_mask[g];
    float4 gx = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m != 0u) {
        const float alpha = aux_alpha[g];
        const float4 gr = grad_x_rgba[g];
        gx.x = ((m & 1u) ? gr.x : 0.f) * alpha;
        gx.y = ((m & 2u) ? gr.y : 0.f) * alpha;
        gx.z = ((m & 4u) ? gr.z : 0.f) * alpha;
    }
    g_out[g] = gx;
}

// pass 2: grad_s[j] = sum over the entries of row j of w_e * g[pixel_e], as a SEGMENTED REDUCTION OVER ENTRIES.
// Real maps have very uneven rows (a surface point seen at a grazing angle by a base view is the neighbour of thousands
// of pixels of other views, most rows of background points have no entry at all): with one lane per row the longest
// rows of a wave set its run time (1.1 ms per 8-view batch; 4.8 ms before zero-weight entries were dropped). Here every
// lane owns ENTRIES instead: a wave takes kSegChunk consecutive entries of the row-sorted list, 64 at a time with
// coalesced index loads and 64 x U independent 16-byte gathers in flight, multiplies, and sums runs of equal row id with
// a wave-wide segmented scan (6 shuffle steps); a run that continues into the next 64 entries is carried in registers.
// Rows that lie inside one chunk are written directly; a row that crosses chunk boundaries leaves one partial record per
// chunk, combined in chunk order by gauss_seg_combine_kernel. Every order is fixed: bitwise reproducible, no atomics,
// and the same for 1 and for C right-hand sides (the multi-RHS slices equal the single-RHS result bit for bit).
constexpr int kSegU = 8;                       // 64-entry steps per wave
constexpr int kSegChunk = 64 * kSegU;          // entries per wave
constexpr int kSegNone = 0x7fffffff;           // row id of a padding lane / "no record"

__device__ __forceinline__ float4 f4_add(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

__device__ __forceinline__ float lane63(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }

// The work of one wave = one chunk `wg` of an index with E entries. key_of[e] names the destination of entry e (equal
// for the entries of one row, entries sorted by it): a row id when `out` is the dense table (stride = Ns), a row ORDINAL
// when `out` is a view's compact row-sum array (stride = its row count; nerfail_gauss_bwd_views).
// PACKED = a per-view index in its compact form (nerfail_gauss_view_pack): `contrib` holds pixel * 2 + (1 if the entry
// starts a row), `row_of` holds ONE int per chunk (the ordinal of the row the chunk's first entry belongs to), and an
// entry's key - its row's ordinal - is that plus the number of row starts up to the entry: 8 bytes per entry instead of 12.
//
// Round 4: a lane owns 8 CONSECUTIVE entries. Round 3 gave lane l the entries l, l + 64, ... and ran a 6-step wave-wide
// segmented scan for every 64 entries: 1 070 vector instructions per wave and chunk, 59 M per 8-view batch - on 16-lane
// SIMDs (a wave64 instruction takes 4 cycles) that alone is 96 of the kernel's 99.6 us (SQ_INSTS_VALU, SQ_WAIT_INST_ANY =
// 42 % of the wave cycles, profiles/r04_k11_counters.txt): the kernel was bound by vector-instruction issue, not by memory.
// Now each lane adds up the runs inside its own 8 entries serially (packed fp32 adds, no cross-lane traffic) and writes the
// rows that begin AND end there directly; what crosses lanes - the run still open at a lane's end, continued through lanes
// without a row start, closed by the first start of a later lane - is ONE segmented scan per 512 entries instead of eight.
// Index and weight loads are two 16-byte loads per lane and array (a wave covers 2 KB contiguous per array).
// Every order is fixed (inside a lane left to right, lanes combined by the scan's fixed tree, chunks by
// seg_combine_chunk in chunk order): bitwise reproducible, no atomics, and identical for 1 and for C right-hand sides.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct V4 { f32x2 lo, hi; };                   // (x, y), (z, w): the sums run on v_pk_add_f32 / v_pk_mul_f32 (same rounding)

__device__ __forceinline__ float4 v4_f4(const V4& v) { return make_float4(v.lo.x, v.lo.y, v.hi.x, v.hi.y); }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i0(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, false); }

// inclusive prefix sum of one int per lane over the 64 lanes (row_shr 1 2 4 8, then the last lane of a row into the rows behind)
__device__ __forceinline__ int wave_incl_sum(int x) {
    x += dpp_i0<0x111, 0xF>(x);
    x += dpp_i0<0x112, 0xF>(x);
    x += dpp_i0<0x114, 0xF>(x);
    x += dpp_i0<0x118, 0xF>(x);
    x += dpp_i0<0x142, 0xA>(x);
    x += dpp_i0<0x143, 0xC>(x);
    return x;
}

// One step of the segmented scan: every lane looks at the lane the DPP control names (a lane without a source - row
// start, masked row - sees key -2, which no entry has) and adds that lane's sums if it holds the same key.
template <int CTRL, int ROW_MASK, bool W4, int C>
__device__ __forceinline__ void seg_scan_step(const int key, V4 (&v)[C]) {
    const int key_src = __builtin_amdgcn_update_dpp(-2, key, CTRL, ROW_MASK, 0xF, false);
    const bool same = key_src == key;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        V4 a;
        a.lo.x = __int_as_float(dpp_i0<CTRL, ROW_MASK>(__float_as_int(v[c].lo.x)));
        a.lo.y = __int_as_float(dpp_i0<CTRL, ROW_MASK>(__float_as_int(v[c].lo.y)));
        a.hi.x = __int_as_float(dpp_i0<CTRL, ROW_MASK>(__float_as_int(v[c].hi.x)));
        a.hi.y = W4 ? __int_as_float(dpp_i0<CTRL, ROW_MASK>(__float_as_int(v[c].hi.y))) : 0.f;   // (the rgb-only step path carries no fourth channel)
        if (same) { v[c].lo += a.lo; v[c].hi += a.hi; }
    }
}

// The state of one lane while it walks its 8 entries left to right.
template <int C>
struct SegLane {
    V4 acc[C];                                  // the run that is open
    V4 H[C];                                    // the part of the ENTERING run (open when the lane begins) that lies in this lane
    bool seen;                                  // a row started in this lane
    int key;                                    // destination of the open run
    __device__ __forceinline__ void init(int kin) {
#pragma unroll
        for (int c = 0; c < C; ++c) { acc[c].lo = acc[c].hi = (f32x2){0.f, 0.f}; H[c] = acc[c]; }
        seen = false;
        key = kin;
    }
    // entry with product p; s: it begins a row; key_after: its destination
    template <class Emit>
    __device__ __forceinline__ void entry(const bool s, const int key_after, const V4 (&p)[C], Emit& emit) {
        if (s) {                                                       // the run open so far ends in front of this entry
            if (!seen) {
#pragma unroll
                for (int c = 0; c < C; ++c) H[c] = acc[c];             // the entering run: finished in seg_finish, with the lanes before
            } else emit(key, acc);                                     // began in this lane: complete
        }
        key = key_after;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const f32x2 lo = acc[c].lo + p[c].lo, hi = acc[c].hi + p[c].hi;
            acc[c].lo = s ? p[c].lo : lo;
            acc[c].hi = s ? p[c].hi : hi;
        }
        seen = seen || s;
    }
};

// After the lanes' own passes: the runs that cross lanes, the chunk's records. Returns the key of the run open at the chunk's
// end (wave-uniform) and whether that run was emitted as a complete row.
template <int C, bool W4, class Emit>
__device__ __forceinline__ int seg_finish(SegLane<C>& st, const int lane, const long wg, const int kin, const bool first_entry_starts,
                                          const int first_row, const bool head_partial, const bool tail_complete,
                                          int* __restrict__ rec_row, float4* __restrict__ rec_val, Emit& emit, bool& tail_emitted) {
    // inclusive segmented scan of the runs still open at the lanes' ends, keyed by the lanes' last keys (equal over a stretch
    // of lanes <=> no row starts in the later ones): Hillis-Steele inside each row of 16 lanes (row_shr 1, 2, 4, 8), then the
    // last lane of a row handed to the next rows (row_bcast:15 / :31) - if that lane's key is mine, everything between it and
    // me has that key too, so its sum is exactly what my run is missing. (__shfl_up is ds_bpermute_b32 on gfx9: DPP moves
    // stay in the vector registers.)
    seg_scan_step<0x111, 0xF, W4>(st.key, st.acc);
    seg_scan_step<0x112, 0xF, W4>(st.key, st.acc);
    seg_scan_step<0x114, 0xF, W4>(st.key, st.acc);
    seg_scan_step<0x118, 0xF, W4>(st.key, st.acc);
    seg_scan_step<0x142, 0xA, W4>(st.key, st.acc);
    seg_scan_step<0x143, 0xC, W4>(st.key, st.acc);
    // the entering run of a lane with a start ends there: (sum over the lanes before = the previous lane's scan value) + H
    {
        V4 tot[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {                                  // wave_shr:1 - lane l receives lane l - 1 (lane 0: nothing, 0)
            V4 x;
            x.lo.x = __int_as_float(dpp_i0<0x138, 0xF>(__float_as_int(st.acc[c].lo.x)));
            x.lo.y = __int_as_float(dpp_i0<0x138, 0xF>(__float_as_int(st.acc[c].lo.y)));
            x.hi.x = __int_as_float(dpp_i0<0x138, 0xF>(__float_as_int(st.acc[c].hi.x)));
            x.hi.y = W4 ? __int_as_float(dpp_i0<0x138, 0xF>(__float_as_int(st.acc[c].hi.y))) : 0.f;
            tot[c].lo = x.lo + st.H[c].lo;
            tot[c].hi = x.hi + st.H[c].hi;
        }
        const bool entering = lane != 0 || !first_entry_starts;        // (the chunk's first entry starting a row closes nothing)
        if (st.seen && entering) {
            if (kin == first_row && head_partial) {                    // began in an earlier chunk: partial (head) record
                rec_row[4 * wg] = kin;
#pragma unroll
                for (int c = 0; c < C; ++c) rec_val[(2 * wg) * C + c] = v4_f4(tot[c]);
            } else emit(kin, tot);
        }
    }
    // the run still open at the end of the chunk
    const int carry_row = __builtin_amdgcn_readlane(st.key, 63);
    V4 carry[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        carry[c].lo = (f32x2){lane63(st.acc[c].lo.x), lane63(st.acc[c].lo.y)};
        carry[c].hi = (f32x2){lane63(st.acc[c].hi.x), W4 ? lane63(st.acc[c].hi.y) : 0.f};
    }
    const bool is_head = carry_row == first_row && head_partial;
    tail_emitted = tail_complete && !is_head;
    if (lane == 0) {
        if (tail_emitted) emit(carry_row, carry);
        else if (tail_complete) {                                      // ends here, began earlier
            rec_row[4 * wg] = carry_row;
#pragma unroll
            for (int c = 0; c < C; ++c) rec_val[(2 * wg) * C + c] = v4_f4(carry[c]);
        } else {                                                       // continues in the next chunk
            rec_row[4 * wg + 1] = carry_row;
            rec_row[4 * wg + 2] = is_head ? 0 : 1;                     // 1: the row STARTS in this chunk
#pragma unroll
            for (int c = 0; c < C; ++c) rec_val[(2 * wg + 1) * C + c] = v4_f4(carry[c]);
        }
    }
    return carry_row;
}

// L (optional, C == 1 and PACKED only): this wave's LDS slice of kSegLdsSlots float4. With it, a FULL chunk takes the
// transposed route: index / weight loads and gathers in the entry-strided pattern (lane l: entries l, l + 64, ... - adjacent
// lanes read adjacent entries, whose pixels share 128-byte lines: ~15 lines per gather instruction instead of ~64 when a lane
// gathers for its own 8 consecutive entries, and the vector-memory pipe processes an instruction line by line), products
// w * g written to LDS entry-major (entry e in slot e ^ ((e >> 3) & 7): a permutation inside every aligned group of 8 slots,
// so the writes stay conflict-free and the read-back - lane l reads slots 8 l + (k ^ (l & 7)) - spreads 8 lanes over all 32
// banks; exactly 8 KB per wave = 5 workgroups per CU) and read back lane-consecutive; complete rows are collected in the same slice by ordinal and leave as one contiguous run of
// full 16-byte-per-lane stores. Same products, same order of additions: the bits equal the direct route's.
constexpr int kSegLdsSlots = kSegChunk;
#ifndef NF_SEG_LDS_MULT
#define NF_SEG_LDS_MULT 1      // (occupancy experiment: 2 halves the waves per CU)
#endif

template <int C, bool PACKED, bool W4 = true>
__device__ __forceinline__ void seg_reduce_chunk(const long wg, const int lane, const long E, const int* __restrict__ row_of,
                                                 const int* __restrict__ contrib, const float* __restrict__ w_sorted,
                                                 const float4* __restrict__ g_pix, const int accumulate,
                                                 float4* __restrict__ grad_spatial, const long Ns,
                                                 int* __restrict__ rec_row, float4* __restrict__ rec_val, float4* L = nullptr) {
    // UB: entries of a lane whose gathers are issued together (register budget: UB * C float4 per lane)
    constexpr int UB = C == 1 ? 8 : (C == 2 ? 4 : (C <= 4 ? 2 : 1));
    static_assert(kSegU == 8, "a lane owns 8 consecutive entries: two 16-byte loads per array");
    const long base = wg * kSegChunk;
    // record slots of this chunk: [4*wg] head row, [4*wg+1] tail row, [4*wg+2] tail-starts-here flag
    if (lane == 0) { rec_row[4 * wg] = kSegNone; rec_row[4 * wg + 1] = kSegNone; rec_row[4 * wg + 2] = 0; }
    if (base >= E) return;                                             // wave-uniform
    const long end = base + kSegChunk < E ? base + kSegChunk : E;
    const bool full = end - base == kSegChunk;                         // wave-uniform: every chunk of a view but its last
    int first_row;
    bool head_partial;                                                 // the chunk's first row began in an earlier chunk
    bool tail_complete;                                                // the row of the chunk's last entry ends with it
    if constexpr (PACKED) {
        first_row = row_of[wg];
        head_partial = base > 0 && (contrib[base] & 1) == 0;
        tail_complete = end >= E || (contrib[end] & 1) != 0;
    } else {
        first_row = row_of[base];
        head_partial = base > 0 && row_of[base - 1] == first_row;
        tail_complete = end >= E || row_of[end] != row_of[end - 1];
    }
    const long e0 = base + (long)lane * kSegU;                         // this lane's first entry
    SegLane<C> st;

    if constexpr (C == 1 && PACKED) {
        if (L != nullptr && full) {
            // ---- A: entry-strided loads, gathers and products -> LDS
            int ids[kSegU];
            float ws[kSegU];
#pragma unroll
            for (int u = 0; u < kSegU; ++u) { ids[u] = contrib[base + u * 64 + lane]; ws[u] = w_sorted[base + u * 64 + lane]; }
            __builtin_amdgcn_sched_barrier(0);
            float4 g[kSegU];
#pragma unroll
            for (int u = 0; u < kSegU; ++u) g[u] = g_pix[ids[u] >> 1];
            // the lane's own 8 consecutive index words (start flags): the lines were just loaded above
            const int4 ia = *reinterpret_cast<const int4*>(contrib + e0), ib = *reinterpret_cast<const int4*>(contrib + e0 + 4);
            __builtin_amdgcn_sched_barrier(0);
            float4* const Lw = L + (lane ^ ((lane >> 3) & 7));
#pragma unroll
            for (int u = 0; u < kSegU; ++u) {
                const f32x2 w2 = {ws[u], ws[u]};
                const f32x2 lo = w2 * (f32x2){g[u].x, g[u].y}, hi = w2 * (f32x2){g[u].z, W4 ? g[u].w : 0.f};
                Lw[64 * u] = make_float4(lo.x, lo.y, hi.x, hi.y);
            }
            const unsigned starts = (unsigned)(ia.x & 1) | (unsigned)(ia.y & 1) << 1 | (unsigned)(ia.z & 1) << 2 | (unsigned)(ia.w & 1) << 3 |
                                    (unsigned)(ib.x & 1) << 4 | (unsigned)(ib.y & 1) << 5 | (unsigned)(ib.z & 1) << 6 | (unsigned)(ib.w & 1) << 7;
            const unsigned counted = lane == 0 ? (starts & ~1u) : starts;
            const int ns = __popc(counted);
            const int kin = first_row + wave_incl_sum(ns) - ns;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (one wave: its LDS operations execute in order)
            // ---- B: the lane's 8 consecutive products back from LDS, then its own pass; rows collected in LDS by ordinal
            float4 pk[kSegU];
            const float4* const Lr = L + 8 * lane;
#pragma unroll
            for (int k = 0; k < kSegU; ++k) pk[k] = Lr[k ^ (lane & 7)];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            auto emit = [&](int row, const V4 (&v)[1]) { L[row - first_row] = v4_f4(v[0]); };
            st.init(kin);
#pragma unroll
            for (int k = 0; k < kSegU; ++k) {
                const bool s = (starts >> k) & 1u;
                V4 p[1];
                p[0].lo = (f32x2){pk[k].x, pk[k].y};
                p[0].hi = (f32x2){pk[k].z, pk[k].w};
                st.entry(s, st.key + ((counted >> k) & 1u), p, emit);
            }
            bool tail_emitted;
            const int carry_row = seg_finish<1, W4>(st, lane, wg, kin, (starts & 1u) != 0u, first_row, head_partial, tail_complete,
                                                    rec_row, rec_val, emit, tail_emitted);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // ---- C: every row between the chunk's first (if it began here) and its last (if it ends here) is complete
            const int lo_key = first_row + (head_partial ? 1 : 0);
            const int hi_key = tail_emitted ? carry_row : carry_row - 1;
            for (int r = lo_key + lane; r <= hi_key; r += 64) grad_spatial[r] = L[r - first_row];
            return;
        }
    }

    // ---- direct route: this lane's 8 entries straight from the arrays (a partial last chunk repeats its last entry and masks
    // the repeats out)
    int id[kSegU], rk[kSegU];
    float w[kSegU];
    unsigned valid = 0xffu;
    if (full) {
        const int4 ia = *reinterpret_cast<const int4*>(contrib + e0), ib = *reinterpret_cast<const int4*>(contrib + e0 + 4);
        const float4 wa = *reinterpret_cast<const float4*>(w_sorted + e0), wb = *reinterpret_cast<const float4*>(w_sorted + e0 + 4);
        id[0] = ia.x; id[1] = ia.y; id[2] = ia.z; id[3] = ia.w; id[4] = ib.x; id[5] = ib.y; id[6] = ib.z; id[7] = ib.w;
        w[0] = wa.x; w[1] = wa.y; w[2] = wa.z; w[3] = wa.w; w[4] = wb.x; w[5] = wb.y; w[6] = wb.z; w[7] = wb.w;
        if constexpr (!PACKED) {
            const int4 ra = *reinterpret_cast<const int4*>(row_of + e0), rb = *reinterpret_cast<const int4*>(row_of + e0 + 4);
            rk[0] = ra.x; rk[1] = ra.y; rk[2] = ra.z; rk[3] = ra.w; rk[4] = rb.x; rk[5] = rb.y; rk[6] = rb.z; rk[7] = rb.w;
        }
    } else {
        valid = 0u;
#pragma unroll
        for (int k = 0; k < kSegU; ++k) {                              // (clamped: unconditional loads)
            const long i = e0 + k;
            const long ic = i < end ? i : end - 1;
            id[k] = contrib[ic];
            w[k] = w_sorted[ic];
            if constexpr (!PACKED) rk[k] = row_of[ic];
            valid |= (i < end ? 1u : 0u) << k;
        }
    }
    // bit k of `starts`: entry k begins a row. The key (destination) of entry k: PACKED - the ordinal first_row + the number
    // of starts among the chunk's entries 1 .. k (the chunk's own first entry does not count: first_row is ITS row);
    // otherwise the row id itself. `kin` = the key of the run that is open when the lane begins (= the previous lane's last key).
    unsigned starts = 0u, counted = 0u;
    int kin;
    if constexpr (PACKED) {
#pragma unroll
        for (int k = 0; k < kSegU; ++k) starts |= (unsigned)(id[k] & 1) << k;
        starts &= valid;
        counted = lane == 0 ? (starts & ~1u) : starts;
        const int ns = __popc(counted);
        kin = first_row + wave_incl_sum(ns) - ns;
    } else {
        const long ep = (e0 < end ? e0 : end) - 1;                     // the entry before this lane's first (>= base - 1)
        const int prev = ep >= 0 ? row_of[ep] : -1;
        kin = lane == 0 ? first_row : prev;
        starts = (lane == 0 ? (head_partial ? 0u : 1u) : (rk[0] != prev ? 1u : 0u));
#pragma unroll
        for (int k = 1; k < kSegU; ++k) starts |= (rk[k] != rk[k - 1] ? 1u : 0u) << k;
        starts &= valid;
    }
    __builtin_amdgcn_sched_barrier(0);

    auto emit = [&](int row, const V4 (&v)[C]) {                       // a row whose sum is complete inside this chunk
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float4 o = v4_f4(v[c]);
            if (accumulate) o = f4_add(grad_spatial[(long)c * Ns + row], o);
            grad_spatial[(long)c * Ns + row] = o;
        }
    };
    st.init(kin);
#pragma unroll
    for (int k0 = 0; k0 < kSegU; k0 += UB) {
        float4 g[UB][C];
        // all gathers of the batch first: UB * C independent 16-byte gathers in flight (the sched_barriers pin the phases;
        // left to itself the compiler chains "gather; s_waitcnt vmcnt(0); use" per entry)
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
            for (int c = 0; c < C; ++c) g[u][c] = g_pix[(long)(id[k0 + u] >> (PACKED ? 1 : 3)) * C + c];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int k = k0 + u;
            const bool s = (starts >> k) & 1u;
            V4 p[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const f32x2 w2 = {w[k], w[k]};
                p[c].lo = w2 * (f32x2){g[u][c].x, g[u][c].y};
                p[c].hi = w2 * (f32x2){g[u][c].z, W4 ? g[u][c].w : 0.f};
                if (!full && !((valid >> k) & 1u)) p[c].lo = p[c].hi = (f32x2){0.f, 0.f};   // (keeps 0 * inf of a repeated entry out)
            }
            st.entry(s, PACKED ? st.key + (int)((counted >> k) & 1u) : rk[k], p, emit);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    bool tail_emitted;
    seg_finish<C, W4>(st, lane, wg, kin, (starts & 1u) != 0u, first_row, head_partial, tail_complete, rec_row, rec_val, emit, tail_emitted);
}

template <int C>
__global__ __launch_bounds__(256) void gauss_seg_reduce_kernel(const int* __restrict__ n_entries, const int* __restrict__ row_of,
                                                               const int* __restrict__ contrib, const float* __restrict__ w_sorted,
                                                               const float4* __restrict__ g_pix, int accumulate,
                                                               float4* __restrict__ grad_spatial, long Ns,
                                                               int* __restrict__ rec_row, float4* __restrict__ rec_val) {
    seg_reduce_chunk<C, false>((long)blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63, *n_entries, row_of, contrib, w_sorted,
                               g_pix, accumulate, grad_spatial, Ns, rec_row, rec_val);
}

// All views of a batch in ONE launch (blockIdx.y = view): every view reduces its own entries into its own compact
// row-sum array, so the views do not touch common memory and need no order among them; gauss_rows_sum_kernel then adds
// the views' sums row by row in view order. (One launch per view, each accumulating into the table, left the chip at
// ~2 waves per SIMD and serialised 16 launches: 0.37 ms per 8-view batch against 0.1 ms here.)
constexpr int kViewsPerLaunch = 16;
struct SegViews {
    long E[kViewsPerLaunch];                    // entries of the view's index
    long chunks[kViewsPerLaunch];               // record slots (waves) of the view
    long n_rows[kViewsPerLaunch];
    long block_start[kViewsPerLaunch];          // first workgroup of the view in the flat list (reduce kernel)
    long total_blocks;
    const int* chunk_ord[kViewsPerLaunch];
    const int* packed[kViewsPerLaunch];
    const float* w_sorted[kViewsPerLaunch];
    const float4* g_pix[kViewsPerLaunch];
    float4* val[kViewsPerLaunch];               // [n_rows] row sums of the view
    int* rec_row[kViewsPerLaunch];
    float4* rec_val[kViewsPerLaunch];
    int nv;
};

// Workgroup -> (view, chunk quad): the launch is the flat list of all views' chunk quads, cut into 8 contiguous parts,
// one per XCD (consecutive workgroup ids go round-robin over the 8 XCDs). Each XCD's L2 then serves ONE view's
// per-pixel gradients (10 MB, swept in step with the rows) instead of all of them at once: with the plain (x = chunk,
// y = view) grid the 16-byte gathers were re-fetched 3.5 times (729 MB of L2 fills per 8-view batch for 205 MB of
// algorithmic bytes, rocprofv3 FETCH_SIZE) and the kernel ran at the fabric's rate.
template <bool W4>
__global__ __launch_bounds__(256) void gauss_seg_reduce_views_kernel(SegViews a) {
    const long per_xcd = (a.total_blocks + 7) >> 3;
    long vb = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((long)(blockIdx.x >> 3) >= per_xcd || vb >= a.total_blocks) return;
    int v = 0;
#pragma unroll
    for (int i = 1; i < kViewsPerLaunch; ++i)
        if (i < a.nv && vb >= a.block_start[i]) v = i;                 // block_start ascends
    const long wg = (vb - a.block_start[v]) * 4 + (threadIdx.x >> 6);
    if (wg >= a.chunks[v]) return;                                     // wave-uniform
    __shared__ float4 lds[4][kSegLdsSlots * NF_SEG_LDS_MULT];                            // one slice per wave, never shared: no barriers
    seg_reduce_chunk<1, true, W4>(wg, threadIdx.x & 63, a.E[v], a.chunk_ord[v], a.packed[v], a.w_sorted[v], a.g_pix[v], 0, a.val[v],
                                  a.n_rows[v], a.rec_row[v], a.rec_val[v], lds[threadIdx.x >> 6]);
}

// ONE view, C right-hand sides (DeepFool's class gradients), over the view's compact index into val[C][n_rows].
template <int C>
__global__ __launch_bounds__(256) void gauss_seg_reduce_packed_kernel(long E, const int* __restrict__ chunk_ord,
                                                                      const int* __restrict__ packed, const float* __restrict__ w_sorted,
                                                                      const float4* __restrict__ g_pix, float4* __restrict__ val,
                                                                      long n_rows, int* __restrict__ rec_row, float4* __restrict__ rec_val) {
    seg_reduce_chunk<C, true>((long)blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63, E, chunk_ord, packed, w_sorted, g_pix, 0,
                              val, n_rows, rec_row, rec_val);
}

// Rows that cross chunk boundaries: the chunk where such a row starts adds up its partial records in chunk order.
template <int C>
__device__ __forceinline__ void seg_combine_chunk(const long k, const int* __restrict__ rec_row, const float4* __restrict__ rec_val,
                                                  const long chunks, const int accumulate, float4* __restrict__ grad_spatial,
                                                  const long Ns) {
    const int row = rec_row[4 * k + 1];
    if (row == kSegNone || rec_row[4 * k + 2] == 0) return;            // no open row here, or it did not start here
    float4 s[C];
#pragma unroll
    for (int c = 0; c < C; ++c) s[c] = rec_val[(2 * k + 1) * C + c];
    for (long kk = k + 1; kk < chunks; ++kk) {
        if (rec_row[4 * kk] == row) {                                  // the row ends in chunk kk
#pragma unroll
            for (int c = 0; c < C; ++c) s[c] = f4_add(s[c], rec_val[(2 * kk) * C + c]);
            break;
        }
        if (rec_row[4 * kk + 1] != row) break;                         // (cannot happen: a row's chunks are contiguous)
#pragma unroll
        for (int c = 0; c < C; ++c) s[c] = f4_add(s[c], rec_val[(2 * kk + 1) * C + c]);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float4 o = s[c];
        if (accumulate) o = f4_add(grad_spatial[(long)c * Ns + row], o);
        grad_spatial[(long)c * Ns + row] = o;
    }
}

template <int C>
__global__ __launch_bounds__(256) void gauss_seg_combine_kernel(const int* __restrict__ rec_row, const float4* __restrict__ rec_val,
                                                                long chunks, int accumulate, float4* __restrict__ grad_spatial,
                                                                long Ns) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= chunks) return;
    seg_combine_chunk<C>(k, rec_row, rec_val, chunks, accumulate, grad_spatial, Ns);
}

__global__ __launch_bounds__(256) void gauss_seg_combine_views_kernel(SegViews a) {
    const int v = blockIdx.y;
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.chunks[v]) return;
    seg_combine_chunk<1>(k, a.rec_row[v], a.rec_val[v], a.chunks[v], 0, a.val[v], a.n_rows[v]);
}

// grad_spatial[j] = (accumulate ? grad_spatial[j] : 0) + sum over the views, IN VIEW ORDER, of the view's sum for row j
// (pos[v][j] = the row's ordinal in view v's compact array, -1 = the view has no entry for it). Ordinals ascend with j,
// so a wave's reads of a view's sums are one nearly contiguous run.
struct RowsSum {
    const int* pos[kViewsPerLaunch];
    const float4* val[kViewsPerLaunch];
    long n_rows[kViewsPerLaunch];               // stride between the right-hand sides of a view's sums
    int nv;
};

// AS:352-392 for one row (igsm_step_rgb_kernel's arithmetic, gauss.hip): the fused epilogue of the rows sum
struct StepArgs {
    const float4* s;
    const float4* s_init;
    float4* out;
    float a, epsilon;
    int targeted;
};

template <int C>
__global__ __launch_bounds__(256) void gauss_rows_sum_kernel(RowsSum a, long Ns, int accumulate, float4* __restrict__ grad_spatial) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Ns) return;
    int p[kViewsPerLaunch];
#pragma unroll
    for (int v = 0; v < kViewsPerLaunch; ++v) p[v] = v < a.nv ? a.pos[v][j] : -1;      // all index loads first
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float4 s = accumulate ? grad_spatial[(long)c * Ns + j] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int v = 0; v < kViewsPerLaunch; ++v)
            if (p[v] >= 0) s = f4_add(s, a.val[v][(long)c * a.n_rows[v] + p[v]]);
        grad_spatial[(long)c * Ns + j] = s;
    }
}

// the same sum written as [Ns,3] (rgb only): the buffer the perturbation-gradient all-reduce moves (23 MB instead of 30.7).
// Round 6, STEP: the NeRFail-S sign step (AS:352-392) as the epilogue - one launch less, and the gradient is neither written nor read
// back (world == 1: nothing is all-reduced); grad3 may be NULL then. 8 us of a 0.265 ms iteration.
// (Measured and dropped in the same round: resolving the rows that cross chunk boundaries HERE - a marker in the unused fourth
// channel of the row sums, the records added up by the thread that needs the row - instead of in gauss_seg_combine_views_kernel
// (8.5 us + a launch). Bit-identical, but only ~0.4 % of the rows are split and still nearly every 64-row WAVE holds one: the
// whole wave walks the record path and this kernel went from 24 to 48 us.)
template <bool STEP>
__global__ __launch_bounds__(256) void gauss_rows_sum3_kernel(RowsSum a, long Ns, int accumulate, float* __restrict__ grad3, StepArgs st) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Ns) return;
    int p[kViewsPerLaunch];
#pragma unroll
    for (int v = 0; v < kViewsPerLaunch; ++v) p[v] = v < a.nv ? a.pos[v][j] : -1;      // all index loads first
    float4 s = accumulate ? make_float4(grad3[3 * j], grad3[3 * j + 1], grad3[3 * j + 2], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int v = 0; v < kViewsPerLaunch; ++v)
        if (p[v] >= 0) s = f4_add(s, a.val[v][p[v]]);
    if (grad3 != nullptr) { grad3[3 * j] = s.x; grad3[3 * j + 1] = s.y; grad3[3 * j + 2] = s.z; }
    if constexpr (STEP) {
        const float4 v = st.s[j], in = st.s_init[j];
        const float sv[3] = {v.x, v.y, v.z}, gv[3] = {s.x, s.y, s.z}, iv[3] = {in.x, in.y, in.z};
        float r[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float sg = (gv[c] > 0.f) ? 1.f : ((gv[c] < 0.f) ? -1.f : 0.f);
            const float stp = __fmul_rn(st.a, sg);
            float q = st.targeted ? __fsub_rn(sv[c], stp) : __fadd_rn(sv[c], stp);
            q = (v.w > 0.f) ? q : 0.f;
            q = fmaxf(q, __fsub_rn(iv[c], st.epsilon));
            q = fminf(q, __fadd_rn(iv[c], st.epsilon));
            r[c] = q;
        }
        st.out[j] = make_float4(r[0], r[1], r[2], v.w);
    }
}

// Row ordinals of a view index: pos[j] = number of non-empty rows before row j, or -1 for an empty row; n_rows = number
// of non-empty rows (flags -> exclusive sum -> fix-up); then the packed entries and the per-chunk first ordinals.
__global__ __launch_bounds__(256) void view_row_flags_kernel(const int* __restrict__ row_ptr, long Ns, int* __restrict__ flags) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < Ns) flags[j] = row_ptr[j + 1] > row_ptr[j] ? 1 : 0;
}

__global__ __launch_bounds__(256) void view_row_pos_kernel(const int* __restrict__ row_ptr, long Ns, int* __restrict__ pos,
                                                           int* __restrict__ n_rows) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Ns) return;
    const bool any = row_ptr[j + 1] > row_ptr[j];
    const int before = pos[j];                                  // the exclusive sum of the flags
    if (j == Ns - 1) *n_rows = before + (any ? 1 : 0);
    pos[j] = any ? before : -1;
}

__global__ __launch_bounds__(256) void view_entry_pack_kernel(const int* __restrict__ row_ptr, long Ns, const int* __restrict__ row_of,
                                                              const int* __restrict__ contrib, const int* __restrict__ pos, long cap,
                                                              int* __restrict__ packed, int* __restrict__ chunk_ord) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= cap || e >= row_ptr[Ns]) return;
    const int row = row_of[e];
    const int start = (e == 0 || row_of[e - 1] != row) ? 1 : 0;
    packed[e] = ((contrib[e] >> 3) << 1) | start;                  // pixel * 2 + "a row starts here"
    if (e % kSegChunk == 0) chunk_ord[e / kSegChunk] = pos[row];
}

// ---- multi-RHS form (DeepFool: the gradients of all class logits of one iteration, deepfool.py:66-96). The index
// arrays are walked ONCE for C right-hand sides; the per-pixel gradients are stored [pixel][C] so that the C float4 a
// contribution needs are one contiguous run (C = 8: one 128-byte line) instead of C scattered 16-byte gathers.
__global__ __launch_bounds__(256) void gauss_pixel_grad_multi_kernel(const float4* __restrict__ ori, const float4* __restrict__ x_saved,
                                                                     const float4* __restrict__ grad_x_rgba, long n, int C,
                                                                     float epsilon, float4* __restrict__ g_out) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;      // (pixel, rhs) pairs, rhs fastest
    if (g >= n * C) return;
    const long pix = g / C;
    const int c_ = (int)(g - pix * C);
    const float4 x = x_saved[pix];
    const float4 o = ori[pix];
    float4 gx = make_float4(0.f, 0.f, 0.f, 0.f);
    if (o.w > 0.f) {
        const float4 gr = grad_x_rgba[(long)c_ * n + pix];
        const float alpha = x.w / 255.0f;
        const float xc[3] = {x.x, x.y, x.z}, oc[3] = {o.x, o.y, o.z}, grc[3] = {gr.x, gr.y, gr.z};
        float gxc[3] = {0.f, 0.f, 0.f};
        float ga = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float d = xc[c] * alpha;
            bool pass = true;
            if (epsilon >= 0.f) {
                pass = (d >= -epsilon) && (d <= epsilon);
                d = fminf(fmaxf(d, -epsilon), epsilon);
            }
            const float pre = oc[c] + d;
            pass = pass && (pre >= 0.f) && (pre <= 255.f);
            const float gd = pass ? grc[c] : 0.f;
            gxc[c] = gd * alpha;
            ga += gd * xc[c];
        }
        gx.x += gxc[0]; gx.y += gxc[1]; gx.z += gxc[2];
        gx.w += ga / 255.0f;
    }
    g_out[g] = gx;
}

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static size_t cub_temp_bytes(long n) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr,
                                       (int*)nullptr, (int)n, 0, 32, (hipStream_t) nullptr);
    return bytes;
}

// Content fingerprint of B equally sized items (32-bit words): out[2b], out[2b+1] = two sums (mod 2^64) of 32-bit hashes of
// (word, position) - every word is mixed NON-LINEARLY with its full position (murmur3 finaliser, two different position
// multipliers) before it is added, so moved, swapped or compensating values change the sums (a plain sum / position-weighted
// sum does not see a swap of two equal-weight positions or +d / -d at positions of equal weight). The sums are order-free
// across threads. The host keys per-view inverted indices on it when a map arrives as an anonymous tensor (a DataLoader hands
// out a fresh tensor every iteration: neither its address nor a version counter says which view it is).
__device__ __forceinline__ unsigned fmix32(unsigned h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ void fp_add(unsigned word, long pos, unsigned long long& s0, unsigned long long& s1) {
    const unsigned p = (unsigned)pos, q = (unsigned)(pos >> 32);
    s0 += fmix32(word ^ fmix32(p * 0x9e3779b1u + q + 1u));
    s1 += fmix32((word + 0x7f4a7c15u) ^ fmix32(p * 0x85ebca77u + q + 0x165667b1u));
}
__global__ __launch_bounds__(256) void fingerprint_kernel(const unsigned* __restrict__ data, long words, unsigned long long* __restrict__ out) {
    const long b = blockIdx.y;
    const unsigned* __restrict__ d = data + b * words;
    unsigned long long s0 = 0, s1 = 0;
    const long quads = ((reinterpret_cast<uintptr_t>(d) & 15) == 0) ? words / 4 : 0;      // 16-byte loads when the item allows
    const uint4* __restrict__ d4 = reinterpret_cast<const uint4*>(d);
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (long)gridDim.x * blockDim.x) {
        const uint4 v = d4[q];
        const long i = 4 * q;
        fp_add(v.x, i, s0, s1); fp_add(v.y, i + 1, s0, s1); fp_add(v.z, i + 2, s0, s1); fp_add(v.w, i + 3, s0, s1);
    }
    for (long i = 4 * quads + (long)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (long)gridDim.x * blockDim.x) fp_add(d[i], i, s0, s1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); }
    __shared__ unsigned long long part[2][4];         // one pair of atomics per WORKGROUP: they all land on 2 addresses per item
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = s0; part[1][threadIdx.x >> 6] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(out + 2 * b, part[0][0] + part[0][1] + part[0][2] + part[0][3]);
        atomicAdd(out + 2 * b + 1, part[1][0] + part[1][1] + part[1][2] + part[1][3]);
    }
}

static long seg_chunks(long B, long P) { return (B * P * 8 + kSegChunk - 1) / kSegChunk; }

// one backward over the inverted index for C right-hand sides; g_pix [B*P][C] float4 is at the start of `scratch`
template <int C>
static int run_seg_reduce(const int32_t* row_ptr, const int32_t* contrib, const float* w_sorted, const int32_t* row_of, long Ns,
                          long B, long P, float* scratch, int accumulate, float* grad_spatial, hipStream_t s) {
    const long chunks = ((seg_chunks(B, P) + 3) / 4) * 4;             // every launched wave owns record slots
    float4* g_pix = (float4*)scratch;
    float4* rec_val = g_pix + (size_t)B * P * C;
    int* rec_row = (int*)(rec_val + (size_t)2 * chunks * C);
    if (!accumulate) {                           // rows without an entry stay 0
        hipError_t e = hipMemsetAsync(grad_spatial, 0, (size_t)C * Ns * 16, s);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync");
    }
    gauss_seg_reduce_kernel<C><<<dim3((unsigned)(chunks / 4)), dim3(256), 0, s>>>(
        row_ptr + Ns, row_of, contrib, w_sorted, g_pix, accumulate, (float4*)grad_spatial, Ns, rec_row, rec_val);
    NF_LAUNCHED("gauss_seg_reduce_kernel");
    gauss_seg_combine_kernel<C><<<dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s>>>(
        rec_row, rec_val, chunks, accumulate, (float4*)grad_spatial, Ns);
    NF_LAUNCHED("gauss_seg_combine_kernel");
    return NERFAIL_OK;
}

}  // namespace nerfail

using namespace nerfail;

extern "C" int nerfail_fingerprint(const void* data, int64_t words_per_item, int64_t n_items, uint64_t* out, void* stream) {
    NF_REQUIRE(words_per_item >= 0 && n_items >= 0 && n_items < 65536, "bad sizes");
    if (n_items == 0) return NERFAIL_OK;
    NF_REQUIRE(data != nullptr && out != nullptr, "NULL pointer");
    hipStream_t s = as_stream(stream);
    hipError_t e = hipMemsetAsync(out, 0, (size_t)n_items * 16, s);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync");
    if (words_per_item == 0) return NERFAIL_OK;
    long bx = (words_per_item / 4 + 255) / 256;      // one 16-byte load per thread and sweep
    if (bx > 256) bx = 256;                          // ~20 sweeps per thread on an 800 x 800 map; the sums are order-free (mod 2^64)
    if (bx < 1) bx = 1;
    fingerprint_kernel<<<dim3((unsigned)bx, (unsigned)n_items), dim3(256), 0, s>>>((const unsigned*)data, words_per_item,
                                                                                  (unsigned long long*)out);
    NF_LAUNCHED("fingerprint_kernel");
    return NERFAIL_OK;
}

extern "C" size_t nerfail_gauss_csr_workspace_bytes(int64_t Ns, int64_t B, int64_t P) {
    if (Ns <= 0 || B <= 0 || P <= 0) return 0;
    const long n = B * P * 8;
    if (n >= (1L << 31) || Ns >= (1L << 31) - 1) return 0;
    return 2 * align256((size_t)n * 4) + align256(cub_temp_bytes(n));
}

extern "C" size_t nerfail_gauss_bwd_scratch_floats(int64_t B, int64_t P, int n_rhs) {
    if (B <= 0 || P <= 0 || n_rhs < 1 || n_rhs > 8) return 0;
    const size_t chunks = (size_t)(((seg_chunks(B, P) + 3) / 4) * 4);
    return (size_t)B * P * 4 * n_rhs + chunks * 2 * 4 * n_rhs + chunks * 4;      // pixel gradients, record values, record rows
}

extern "C" int nerfail_gauss_csr_build(const float* weight_and_index, int64_t Ns, int64_t B, int64_t P, int32_t* row_ptr,
                                       int32_t* contrib, float* w_sorted, int32_t* row_of, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    NF_REQUIRE(Ns > 0 && B > 0 && P > 0, "bad sizes");
    const long n = B * P * 8;
    NF_REQUIRE(n < (1L << 31) && Ns < (1L << 31) - 1, "batch too large for 32-bit CSR ids");
    NF_REQUIRE(weight_and_index && row_ptr && contrib && w_sorted && row_of && workspace, "NULL pointer");
    const size_t need = nerfail_gauss_csr_workspace_bytes(Ns, B, P);
    NF_REQUIRE(workspace_bytes >= need, "workspace too small (nerfail_gauss_csr_workspace_bytes)");
    hipStream_t s = as_stream(stream);
    char* ws = (char*)workspace;
    const size_t seg = align256((size_t)n * 4);
    unsigned* keys_in = (unsigned*)ws;
    unsigned* keys_out = (unsigned*)row_of;       // the sorted keys ARE the row of every entry (Ns = dropped, at the end)
    int* vals_in = (int*)(ws + seg);
    void* temp = ws + 2 * seg;
    size_t temp_bytes = workspace_bytes - 2 * seg;
    csr_keys_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(weight_and_index, Ns, B, P, keys_in, vals_in);
    NF_LAUNCHED("csr_keys_kernel");
    int bits = 1;
    while ((1L << bits) < Ns + 2) ++bits;         // sort only the significant key bits (stable LSD radix); key Ns = dropped
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, vals_in, contrib, (int)n, 0, bits, s);
    if (e != hipSuccess) return hip_fail(e, "hipcub::DeviceRadixSort::SortPairs");
    const long rows = Ns;
    csr_rows_kernel<<<dim3((unsigned)((rows + 1 + 255) / 256)), dim3(256), 0, s>>>(keys_out, n, rows, row_ptr);
    NF_LAUNCHED("csr_rows_kernel");
    csr_weights_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(weight_and_index, contrib, n, P, w_sorted);
    NF_LAUNCHED("csr_weights_kernel");
    return NERFAIL_OK;
}

extern "C" int nerfail_gauss_bwd_csr(const float* ori_img, const float* x, const float* grad_x, const float* grad_x_rgba,
                                     const int32_t* row_ptr, const int32_t* contrib, const float* w_sorted,
                                     const int32_t* row_of, int64_t Ns, int64_t B, int64_t P, float epsilon, float* scratch,
                                     int accumulate, float* grad_spatial, void* stream) {
    NF_REQUIRE(Ns > 0 && B > 0 && P > 0, "bad sizes");
    NF_REQUIRE(ori_img && x && row_ptr && contrib && w_sorted && row_of && scratch && grad_spatial, "NULL pointer");
    hipStream_t s = as_stream(stream);
    const long n = B * P;
    gauss_pixel_grad_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(
        (const float4*)ori_img, (const float4*)x, (const float4*)grad_x, (const float4*)grad_x_rgba, n, epsilon,
        (float4*)scratch);
    NF_LAUNCHED("gauss_pixel_grad_kernel");
    return run_seg_reduce<1>(row_ptr, contrib, w_sorted, row_of, Ns, B, P, scratch, accumulate, grad_spatial, s);
}

static long view_chunks(long n_entries) { return ((n_entries + kSegChunk - 1) / kSegChunk + 3) / 4 * 4; }

extern "C" size_t nerfail_gauss_view_pack_workspace_bytes(int64_t Ns) {
    if (Ns <= 0 || Ns >= (1L << 31) - 1) return 0;
    size_t bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const int*)nullptr, (int*)nullptr, (int)Ns, (hipStream_t) nullptr);
    return align256((size_t)Ns * 4) + align256(bytes);
}

extern "C" int64_t nerfail_gauss_view_chunks(int64_t n_entries) { return n_entries < 0 ? 0 : view_chunks(n_entries); }

extern "C" int nerfail_gauss_view_pack(const int32_t* row_ptr, const int32_t* row_of, const int32_t* contrib, int64_t Ns,
                                       int64_t entry_capacity, int32_t* pos, int32_t* packed, int32_t* chunk_ord,
                                       int32_t* n_rows, void* workspace, size_t workspace_bytes, void* stream) {
    NF_REQUIRE(Ns > 0 && Ns < (1L << 31) - 1 && entry_capacity >= 0, "bad sizes");
    NF_REQUIRE(row_ptr && row_of && contrib && pos && packed && chunk_ord && n_rows && workspace, "NULL pointer");
    NF_REQUIRE(workspace_bytes >= nerfail_gauss_view_pack_workspace_bytes(Ns), "workspace too small (nerfail_gauss_view_pack_workspace_bytes)");
    hipStream_t s = as_stream(stream);
    int* flags = (int*)workspace;
    void* temp = (char*)workspace + align256((size_t)Ns * 4);
    size_t temp_bytes = workspace_bytes - align256((size_t)Ns * 4);
    const unsigned gb = (unsigned)((Ns + 255) / 256);
    view_row_flags_kernel<<<dim3(gb), dim3(256), 0, s>>>(row_ptr, Ns, flags);
    NF_LAUNCHED("view_row_flags_kernel");
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(temp, temp_bytes, flags, pos, (int)Ns, s);
    if (e != hipSuccess) return hip_fail(e, "hipcub::DeviceScan::ExclusiveSum");
    view_row_pos_kernel<<<dim3(gb), dim3(256), 0, s>>>(row_ptr, Ns, pos, n_rows);
    NF_LAUNCHED("view_row_pos_kernel");
    if (entry_capacity > 0) {
        view_entry_pack_kernel<<<dim3((unsigned)((entry_capacity + 255) / 256)), dim3(256), 0, s>>>(row_ptr, Ns, row_of, contrib, pos,
                                                                                                   entry_capacity, packed, chunk_ord);
        NF_LAUNCHED("view_entry_pack_kernel");
    }
    return NERFAIL_OK;
}

static bool view_ok(const nerfail_view_index& v, long Ns, long P) {
    return v.packed && v.w_sorted && v.chunk_ord && v.pos && v.n_entries >= 0 && v.n_entries <= 8 * P && v.n_rows >= 0 &&
           v.n_rows <= v.n_entries && v.n_rows <= Ns && (v.n_entries == 0) == (v.n_rows == 0);
}

extern "C" size_t nerfail_gauss_bwd_views_scratch_floats(const nerfail_view_index* views, int n_views, int64_t P, int n_rhs) {
    if (views == nullptr || n_views < 1 || P <= 0 || n_rhs < 1 || n_rhs > 8) return 0;
    size_t f = (size_t)n_views * P * 4 * n_rhs;                         // per-pixel gradients of the batch
    for (int v = 0; v < n_views; ++v) {
        if (views[v].n_entries < 0 || views[v].n_rows < 0) return 0;
        const size_t chunks = (size_t)view_chunks(views[v].n_entries);
        f += chunks * 2 * 4 * n_rhs + chunks * 4 + (size_t)views[v].n_rows * 4 * n_rhs;   // record values, record rows, row sums
    }
    return f;
}

// steps 2 + 3 of the batched backward: every view's entries -> its own row sums (one launch for up to 16 views), then the
// views' sums added per row in view order, into [Ns,4] (grad4) or [Ns,3] (grad3)
static int reduce_views(const nerfail_view_index* views, int n_views, long Ns, long P, float* scratch, float* grad4, float* grad3,
                        hipStream_t s, const StepArgs* step = nullptr) {
    const bool rgb = grad3 != nullptr || step != nullptr;
    const long n = (long)n_views * P;
    float* cursor = scratch + (size_t)n * 4;
    for (int v0 = 0; v0 < n_views; v0 += kViewsPerLaunch) {
        SegViews a;
        RowsSum r;
        const int nv = n_views - v0 < kViewsPerLaunch ? n_views - v0 : kViewsPerLaunch;
        a.nv = r.nv = nv;
        long max_chunks = 0;
        for (int i = 0; i < kViewsPerLaunch; ++i) {
            const nerfail_view_index& vi = views[v0 + (i < nv ? i : 0)];     // (unused slots repeat slot 0: valid pointers)
            const long chunks = i < nv ? view_chunks(vi.n_entries) : 0;
            a.E[i] = vi.n_entries; a.chunks[i] = chunks; a.n_rows[i] = r.n_rows[i] = vi.n_rows;
            a.chunk_ord[i] = vi.chunk_ord; a.packed[i] = vi.packed; a.w_sorted[i] = vi.w_sorted;
            a.g_pix[i] = (const float4*)scratch + (size_t)(v0 + (i < nv ? i : 0)) * P;
            a.rec_val[i] = (float4*)cursor;
            a.rec_row[i] = (int*)(cursor + (size_t)chunks * 8);
            a.val[i] = (float4*)(cursor + (size_t)chunks * 12);
            r.pos[i] = vi.pos; r.val[i] = a.val[i];
            if (i < nv) cursor += (size_t)chunks * 12 + (size_t)vi.n_rows * 4;
            if (chunks > max_chunks) max_chunks = chunks;
        }
        a.total_blocks = 0;
        for (int i = 0; i < kViewsPerLaunch; ++i) { a.block_start[i] = a.total_blocks; a.total_blocks += a.chunks[i] / 4; }
        if (max_chunks > 0) {
            if (rgb) gauss_seg_reduce_views_kernel<false><<<dim3((unsigned)(((a.total_blocks + 7) / 8) * 8)), dim3(256), 0, s>>>(a);
            else gauss_seg_reduce_views_kernel<true><<<dim3((unsigned)(((a.total_blocks + 7) / 8) * 8)), dim3(256), 0, s>>>(a);
            NF_LAUNCHED("gauss_seg_reduce_views_kernel");
            gauss_seg_combine_views_kernel<<<dim3((unsigned)((max_chunks + 255) / 256), (unsigned)nv), dim3(256), 0, s>>>(a);
            NF_LAUNCHED("gauss_seg_combine_views_kernel");
        }
        if (rgb) {
            const bool last = v0 + kViewsPerLaunch >= n_views;
            if (step != nullptr && last) {
                gauss_rows_sum3_kernel<true><<<dim3((unsigned)((Ns + 255) / 256)), dim3(256), 0, s>>>(r, Ns, v0 > 0 ? 1 : 0, grad3, *step);
            } else {
                gauss_rows_sum3_kernel<false><<<dim3((unsigned)((Ns + 255) / 256)), dim3(256), 0, s>>>(r, Ns, v0 > 0 ? 1 : 0, grad3, StepArgs{});
            }
            NF_LAUNCHED("gauss_rows_sum3_kernel");
        } else {
            gauss_rows_sum_kernel<1><<<dim3((unsigned)((Ns + 255) / 256)), dim3(256), 0, s>>>(r, Ns, v0 > 0 ? 1 : 0, (float4*)grad4);
            NF_LAUNCHED("gauss_rows_sum_kernel");
        }
    }
    return NERFAIL_OK;
}

extern "C" int nerfail_gauss_bwd_views(const float* ori_img, const float* x, const float* grad_x, const float* grad_x_rgba,
                                       const nerfail_view_index* views, int n_views, int64_t Ns, int64_t P, float epsilon,
                                       float* scratch, float* grad_spatial, void* stream) {
    NF_REQUIRE(Ns > 0 && P > 0 && n_views >= 1, "bad sizes");
    NF_REQUIRE(ori_img && x && views && scratch && grad_spatial, "NULL pointer");
    for (int v = 0; v < n_views; ++v)
        NF_REQUIRE(view_ok(views[v], Ns, P), "a view index is incomplete or inconsistent (NULL array, n_entries > 8 P, n_rows > n_entries)");
    hipStream_t s = as_stream(stream);
    const long n = (long)n_views * P;
    // 1. per-pixel gradients of the whole batch in one launch
    gauss_pixel_grad_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(
        (const float4*)ori_img, (const float4*)x, (const float4*)grad_x, (const float4*)grad_x_rgba, n, epsilon,
        (float4*)scratch);
    NF_LAUNCHED("gauss_pixel_grad_kernel");
    return reduce_views(views, n_views, Ns, P, scratch, grad_spatial, nullptr, s);
}

extern "C" int nerfail_gauss_bwd_views_rgb(const float* aux_alpha, const unsigned char* aux_mask, const float* grad_x_rgba,
                                           const nerfail_view_index* views, int n_views, int64_t Ns, int64_t P, float* scratch,
                                           float* grad_rgb, void* stream) {
    NF_REQUIRE(Ns > 0 && P > 0 && n_views >= 1, "bad sizes");
    NF_REQUIRE(aux_alpha && aux_mask && grad_x_rgba && views && scratch && grad_rgb, "NULL pointer");
    for (int v = 0; v < n_views; ++v)
        NF_REQUIRE(view_ok(views[v], Ns, P), "a view index is incomplete or inconsistent (NULL array, n_entries > 8 P, n_rows > n_entries)");
    hipStream_t s = as_stream(stream);
    const long n = (long)n_views * P;
    gauss_pixel_grad_rgb_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(aux_alpha, aux_mask, (const float4*)grad_x_rgba, n,
                                                                                        (float4*)scratch);
    NF_LAUNCHED("gauss_pixel_grad_rgb_kernel");
    return reduce_views(views, n_views, Ns, P, scratch, nullptr, grad_rgb, s);
}

// nerfail_gauss_bwd_views_rgb with the NeRFail-S sign step (nerfail_igsm_step_rgb) as the epilogue of its last launch
extern "C" int nerfail_gauss_bwd_views_rgb_step(const float* aux_alpha, const unsigned char* aux_mask, const float* grad_x_rgba,
                                                const nerfail_view_index* views, int n_views, int64_t Ns, int64_t P, float* scratch,
                                                float* grad_rgb, const float* spatial, const float* spatial_init, float a, float epsilon,
                                                int targeted, float* spatial_out, void* stream) {
    NF_REQUIRE(Ns > 0 && P > 0 && n_views >= 1, "bad sizes");
    NF_REQUIRE(aux_alpha && aux_mask && grad_x_rgba && views && scratch && spatial && spatial_init && spatial_out, "NULL pointer");
    NF_REQUIRE(n_views <= kViewsPerLaunch || grad_rgb != nullptr, "more views than one launch sums need grad_rgb as the running sum");
    NF_REQUIRE(spatial_out != spatial && spatial_out != spatial_init, "spatial_out must not alias an input");
    for (int v = 0; v < n_views; ++v)
        NF_REQUIRE(view_ok(views[v], Ns, P), "a view index is incomplete or inconsistent (NULL array, n_entries > 8 P, n_rows > n_entries)");
    hipStream_t s = as_stream(stream);
    const long n = (long)n_views * P;
    gauss_pixel_grad_rgb_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(aux_alpha, aux_mask, (const float4*)grad_x_rgba, n,
                                                                                        (float4*)scratch);
    NF_LAUNCHED("gauss_pixel_grad_rgb_kernel");
    StepArgs st;
    st.s = (const float4*)spatial; st.s_init = (const float4*)spatial_init; st.out = (float4*)spatial_out;
    st.a = a; st.epsilon = epsilon; st.targeted = targeted;
    return reduce_views(views, n_views, Ns, P, scratch, nullptr, grad_rgb, s, &st);
}

// ONE view, C right-hand sides over the view's compact index: reduce -> combine -> expand through pos
template <int C>
static int run_view_multi(const nerfail_view_index& vi, long Ns, long P, float* scratch, float* grad_spatial, hipStream_t s) {
    const long chunks = view_chunks(vi.n_entries);
    const float4* g_pix = (const float4*)scratch;
    float* cursor = scratch + (size_t)P * 4 * C;
    float4* rec_val = (float4*)cursor;
    int* rec_row = (int*)(cursor + (size_t)chunks * 8 * C);
    float4* val = (float4*)(cursor + (size_t)chunks * 8 * C + (size_t)chunks * 4);
    if (chunks > 0) {
        gauss_seg_reduce_packed_kernel<C><<<dim3((unsigned)(chunks / 4)), dim3(256), 0, s>>>(
            vi.n_entries, vi.chunk_ord, vi.packed, vi.w_sorted, g_pix, val, vi.n_rows, rec_row, rec_val);
        NF_LAUNCHED("gauss_seg_reduce_packed_kernel");
        gauss_seg_combine_kernel<C><<<dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s>>>(rec_row, rec_val, chunks, 0, val, vi.n_rows);
        NF_LAUNCHED("gauss_seg_combine_kernel");
    }
    RowsSum r;
    r.nv = 1;
    for (int i = 0; i < kViewsPerLaunch; ++i) { r.pos[i] = vi.pos; r.val[i] = val; r.n_rows[i] = vi.n_rows; }
    gauss_rows_sum_kernel<C><<<dim3((unsigned)((Ns + 255) / 256)), dim3(256), 0, s>>>(r, Ns, 0, (float4*)grad_spatial);
    NF_LAUNCHED("gauss_rows_sum_kernel");
    return NERFAIL_OK;
}

extern "C" int nerfail_gauss_bwd_view_multi(const float* ori_img, const float* x, const float* grad_x_rgba, int n_rhs,
                                            const nerfail_view_index* view, int64_t Ns, int64_t P, float epsilon,
                                            float* scratch, float* grad_spatial, void* stream) {
    NF_REQUIRE(Ns > 0 && P > 0, "bad sizes");
    NF_REQUIRE(n_rhs >= 1 && n_rhs <= 8, "n_rhs must be in 1..8");
    NF_REQUIRE(ori_img && x && grad_x_rgba && view && scratch && grad_spatial, "NULL pointer");
    NF_REQUIRE(view_ok(*view, Ns, P), "the view index is incomplete or inconsistent");
    hipStream_t s = as_stream(stream);
    const long total = P * n_rhs;
    gauss_pixel_grad_multi_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(
        (const float4*)ori_img, (const float4*)x, (const float4*)grad_x_rgba, P, n_rhs, epsilon, (float4*)scratch);
    NF_LAUNCHED("gauss_pixel_grad_multi_kernel");
    switch (n_rhs) {
        case 1: return run_view_multi<1>(*view, Ns, P, scratch, grad_spatial, s);
        case 2: return run_view_multi<2>(*view, Ns, P, scratch, grad_spatial, s);
        case 3: return run_view_multi<3>(*view, Ns, P, scratch, grad_spatial, s);
        case 4: return run_view_multi<4>(*view, Ns, P, scratch, grad_spatial, s);
        case 5: return run_view_multi<5>(*view, Ns, P, scratch, grad_spatial, s);
        case 6: return run_view_multi<6>(*view, Ns, P, scratch, grad_spatial, s);
        case 7: return run_view_multi<7>(*view, Ns, P, scratch, grad_spatial, s);
        default: return run_view_multi<8>(*view, Ns, P, scratch, grad_spatial, s);
    }
}

extern "C" int nerfail_gauss_bwd_csr_multi(const float* ori_img, const float* x, const float* grad_x_rgba, int n_rhs,
                                           const int32_t* row_ptr, const int32_t* contrib, const float* w_sorted,
                                           const int32_t* row_of, int64_t Ns, int64_t B, int64_t P, float epsilon,
                                           float* scratch, float* grad_spatial, void* stream) {
    NF_REQUIRE(Ns > 0 && B > 0 && P > 0, "bad sizes");
    NF_REQUIRE(n_rhs >= 1 && n_rhs <= 8, "n_rhs must be in 1..8");
    NF_REQUIRE(ori_img && x && grad_x_rgba && row_ptr && contrib && w_sorted && row_of && scratch && grad_spatial, "NULL pointer");
    hipStream_t s = as_stream(stream);
    const long n = B * P;
    const long total = n * n_rhs;
    gauss_pixel_grad_multi_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(
        (const float4*)ori_img, (const float4*)x, (const float4*)grad_x_rgba, n, n_rhs, epsilon, (float4*)scratch);
    NF_LAUNCHED("gauss_pixel_grad_multi_kernel");
    switch (n_rhs) {
        case 1: return run_seg_reduce<1>(row_ptr, contrib, w_sorted, row_of, Ns, B, P, scratch, 0, grad_spatial, s);
        case 2: return run_seg_reduce<2>(row_ptr, contrib, w_sorted, row_of, Ns, B, P, scratch, 0, grad_spatial, s);
        case 3: return run_seg_reduce<3>(row_ptr, contrib, w_sorted, row_of, Ns, B, P, scratch, 0, grad_spatial, s);
        case 4: return run_seg_reduce<4>(row_ptr, contrib, w_sorted, row_of, Ns, B, P, scratch, 0, grad_spatial, s);
        case 5: return run_seg_reduce<5>(row_ptr, contrib, w_sorted, row_of, Ns, B, P, scratch, 0, grad_spatial, s);
        case 6: return run_seg_reduce<6>(row_ptr, contrib, w_sorted, row_of, Ns, B, P, scratch, 0, grad_spatial, s);
        case 7: return run_seg_reduce<7>(row_ptr, contrib, w_sorted, row_of, Ns, B, P, scratch, 0, grad_spatial, s);
        default: return run_seg_reduce<8>(row_ptr, contrib, w_sorted, row_of, Ns, B, P, scratch, 0, grad_spatial, s);
    }
}
