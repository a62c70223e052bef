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

// One step of the segmented scan: every lane looks at the lane the DPP control names (a lane without a source - row
// start, masked row - sees key -2, which no entry has) and adds that lane's sums if it holds the same key.
template <int CTRL, int ROW_MASK = 0xF, bool W4 = true, int C>
__device__ __forceinline__ void seg_scan_step(const int key, float4 (&v)[C]) {
    const int key_src = __builtin_amdgcn_update_dpp(-2, key, CTRL, ROW_MASK, 0xF, false);
    const bool same = key_src == key;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float ax = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[c].x), CTRL, ROW_MASK, 0xF, false));
        const float ay = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[c].y), CTRL, ROW_MASK, 0xF, false));
        const float az = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[c].z), CTRL, ROW_MASK, 0xF, false));
        if (same) { v[c].x += ax; v[c].y += ay; v[c].z += az; }
        if constexpr (W4) {                                             // (the rgb-only step path carries no fourth channel)
            const float aw = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[c].w), CTRL, ROW_MASK, 0xF, false));
            if (same) v[c].w += aw;
        }
    }
}

// The work of one wave = one chunk `wg` of an index with E entries. key_of[e] names the destination of entry e (equal
// for the entries of one row, entries sorted by it): a row id when `out` is the dense table (stride = Ns), a row ORDINAL
// when `out` is a view's compact row-sum array (stride = its row count; nerfail_gauss_bwd_views).
// PACKED = a per-view index in its compact form (nerfail_gauss_view_pack): `contrib` holds pixel * 2 + (1 if the entry
// starts a row), `row_of` holds ONE int per chunk (the ordinal of the row the chunk's first entry belongs to), and an
// entry's key - its row's ordinal - is that plus the number of row starts up to the entry (a ballot and a bit count per
// 64 entries): 8 bytes per entry instead of 12.
template <int C, bool PACKED, bool W4 = true>
__device__ __forceinline__ void seg_reduce_chunk(const long wg, const int lane, const long E, const int* __restrict__ row_of,
                                                 const int* __restrict__ contrib, const float* __restrict__ w_sorted,
                                                 const float4* __restrict__ g_pix, const int accumulate,
                                                 float4* __restrict__ grad_spatial, const long Ns,
                                                 int* __restrict__ rec_row, float4* __restrict__ rec_val) {
    // UB: steps whose gathers are issued together (register budget: UB * C float4 per lane)
    constexpr int UB = C == 1 ? 8 : (C == 2 ? 4 : (C <= 4 ? 2 : 1));
    const long base = wg * kSegChunk;
    // record slots of this chunk: [4*wg] head row, [4*wg+1] tail row, [4*wg+2] tail-starts-here flag
    if (lane == 0) { rec_row[4 * wg] = kSegNone; rec_row[4 * wg + 1] = kSegNone; rec_row[4 * wg + 2] = 0; }
    if (base >= E) return;                                             // wave-uniform
    const long end = base + kSegChunk < E ? base + kSegChunk : E;
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
    int running = first_row;                                           // PACKED: key of the last entry handled so far
    int carry_row = kSegNone;
    float4 carry[C];
#pragma unroll
    for (int c = 0; c < C; ++c) carry[c] = make_float4(0.f, 0.f, 0.f, 0.f);

    auto emit = [&](int row, const float4 (&v)[C], bool owner) {       // a row whose sum is complete inside this chunk
        if (!owner) return;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float4 o = v[c];
            if (accumulate) o = f4_add(grad_spatial[(long)c * Ns + row], o);
            grad_spatial[(long)c * Ns + row] = o;
        }
    };

#pragma unroll 1
    for (int u0 = 0; u0 < kSegU; u0 += UB) {
        if (base + (long)u0 * 64 >= end) break;                        // wave-uniform
        int key[UB], id[UB];
        float w[UB];
        float4 g[UB][C];
        // all index loads of the batch first, then all gathers: UB * C independent 16-byte gathers in flight. (Left to
        // itself the compiler interleaves "load contrib[u]; s_waitcnt vmcnt(0); gather" per u - 2 * UB memory round
        // trips in a chain, each wait also draining the gather before it; the sched_barriers pin the two phases.)
#pragma unroll
        for (int u = 0; u < UB; ++u) {                                 // (clamped: unconditional loads)
            const long i = base + (long)(u0 + u) * 64 + lane;
            const long ic = i < end ? i : end - 1;
            id[u] = contrib[ic];
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const long i = base + (long)(u0 + u) * 64 + lane;
            const bool ok = i < end;
            const long ic = ok ? i : end - 1;
            if constexpr (PACKED) key[u] = ok ? 0 : kSegNone;          // (the ordinal is counted below, step by step)
            else key[u] = ok ? row_of[ic] : kSegNone;
            w[u] = ok ? w_sorted[ic] : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
            for (int c = 0; c < C; ++c) g[u][c] = g_pix[(long)(id[u] >> (PACKED ? 1 : 3)) * C + c];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (base + (long)(u0 + u) * 64 >= end) break;              // wave-uniform
            if constexpr (PACKED) {
                // row starts among these 64 entries (the chunk's very first entry does not count: first_row is ITS row)
                const bool st = key[u] != kSegNone && (id[u] & 1) != 0 && !(u0 + u == 0 && lane == 0);
                const unsigned long long m = __ballot(st);
                const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                if (key[u] != kSegNone) key[u] = running + below + (st ? 1 : 0);
                running += __popcll(m);
            }
            float4 v[C];
#pragma unroll
            for (int c = 0; c < C; ++c)
                v[c] = key[u] != kSegNone ? make_float4(w[u] * g[u][c].x, w[u] * g[u][c].y, w[u] * g[u][c].z, W4 ? w[u] * g[u][c].w : 0.f)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);   // (keeps 0 * inf of a clamped lane out)
            // inclusive segmented scan over the 64 lanes (keys are sorted: equal keys are contiguous), on DPP moves:
            // Hillis-Steele inside each row of 16 lanes (row_shr 1, 2, 4, 8), then the last lane of a row handed to the
            // next row (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) - if that lane's key is mine,
            // everything between it and me has that key too, so its sum is exactly what my run is missing.
            // (__shfl_up is ds_bpermute_b32 on gfx9: 36 trips through the CU's LDS crossbar per 64 entries kept this
            // kernel at 186 us per 8-view batch whatever its HBM traffic was.)
            seg_scan_step<0x111, 0xF, W4>(key[u], v);
            seg_scan_step<0x112, 0xF, W4>(key[u], v);
            seg_scan_step<0x114, 0xF, W4>(key[u], v);
            seg_scan_step<0x118, 0xF, W4>(key[u], v);
            seg_scan_step<0x142, 0xA, W4>(key[u], v);
            seg_scan_step<0x143, 0xC, W4>(key[u], v);
            const int key0 = __builtin_amdgcn_readlane(key[u], 0);
            if (key[u] == key0 && key0 == carry_row) {                 // the run carried over from the previous 64 entries
#pragma unroll
                for (int c = 0; c < C; ++c) v[c] = f4_add(carry[c], v[c]);
            } else if (lane == 0 && carry_row != kSegNone && carry_row != key0) {
                // the carried run ended exactly at the 64-entry boundary: it is complete now
                const bool is_head = carry_row == first_row && head_partial;
                if (is_head) {
                    rec_row[4 * wg] = carry_row;
#pragma unroll
                    for (int c = 0; c < C; ++c) rec_val[(2 * wg) * C + c] = carry[c];
                } else emit(carry_row, carry, true);
            }
            const int key_next = __builtin_amdgcn_update_dpp(kSegNone, key[u], 0x130, 0xF, 0xF, false);   // wave_shl:1 = lane + 1
            const bool closed = lane != 63 && key_next != key[u] && key[u] != kSegNone;   // run ends inside these 64 entries
            if (closed) {
                if (key[u] == first_row && head_partial) {             // began in an earlier chunk: partial (head) record
                    rec_row[4 * wg] = key[u];
#pragma unroll
                    for (int c = 0; c < C; ++c) rec_val[(2 * wg) * C + c] = v[c];
                } else emit(key[u], v, true);
            }
            carry_row = __builtin_amdgcn_readlane(key[u], 63);         // the run of the last lane stays open
#pragma unroll
            for (int c = 0; c < C; ++c)
                carry[c] = make_float4(lane63(v[c].x), lane63(v[c].y), lane63(v[c].z), W4 ? lane63(v[c].w) : 0.f);
        }
    }
    if (lane == 0 && carry_row != kSegNone) {                          // the run still open at the end of the chunk
        const bool complete = tail_complete;
        const bool is_head = carry_row == first_row && head_partial;
        if (complete && !is_head) emit(carry_row, carry, true);
        else if (complete) {                                           // ends here, began earlier
            rec_row[4 * wg] = carry_row;
#pragma unroll
            for (int c = 0; c < C; ++c) rec_val[(2 * wg) * C + c] = carry[c];
        } else {                                                       // continues in the next chunk
            rec_row[4 * wg + 1] = carry_row;
            rec_row[4 * wg + 2] = is_head ? 0 : 1;                     // 1: the row STARTS in this chunk
#pragma unroll
            for (int c = 0; c < C; ++c) rec_val[(2 * wg + 1) * C + c] = carry[c];
        }
    }
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
    seg_reduce_chunk<1, true, W4>(wg, threadIdx.x & 63, a.E[v], a.chunk_ord[v], a.packed[v], a.w_sorted[v], a.g_pix[v], 0, a.val[v],
                                  a.n_rows[v], a.rec_row[v], a.rec_val[v]);
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

// the same sum written as [Ns,3] (rgb only): the buffer the perturbation-gradient all-reduce moves (23 MB instead of 30.7)
__global__ __launch_bounds__(256) void gauss_rows_sum3_kernel(RowsSum a, long Ns, int accumulate, float* __restrict__ grad3) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Ns) return;
    int p[kViewsPerLaunch];
#pragma unroll
    for (int v = 0; v < kViewsPerLaunch; ++v) p[v] = v < a.nv ? a.pos[v][j] : -1;      // all index loads first
    float4 s = accumulate ? make_float4(grad3[3 * j], grad3[3 * j + 1], grad3[3 * j + 2], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int v = 0; v < kViewsPerLaunch; ++v)
        if (p[v] >= 0) s = f4_add(s, a.val[v][p[v]]);
    grad3[3 * j] = s.x; grad3[3 * j + 1] = s.y; grad3[3 * j + 2] = s.z;
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
                        hipStream_t s) {
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
            if (grad3 != nullptr) gauss_seg_reduce_views_kernel<false><<<dim3((unsigned)(((a.total_blocks + 7) / 8) * 8)), dim3(256), 0, s>>>(a);
            else gauss_seg_reduce_views_kernel<true><<<dim3((unsigned)(((a.total_blocks + 7) / 8) * 8)), dim3(256), 0, s>>>(a);
            NF_LAUNCHED("gauss_seg_reduce_views_kernel");
            gauss_seg_combine_views_kernel<<<dim3((unsigned)((max_chunks + 255) / 256), (unsigned)nv), dim3(256), 0, s>>>(a);
            NF_LAUNCHED("gauss_seg_combine_views_kernel");
        }
        if (grad3 != nullptr) {
            gauss_rows_sum3_kernel<<<dim3((unsigned)((Ns + 255) / 256)), dim3(256), 0, s>>>(r, Ns, v0 > 0 ? 1 : 0, grad3);
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
