// K8, accelerated: exact 8-nearest-neighbour build over a uniform grid (same result, bit for bit, as the brute-force
// kernel in knn.hip and as oracle/knn.py). Replaces create_index_and_dist.py:126-145.
//
//   build:  bounding box of the point set -> G^3 cells (G ~ M^(1/3)) -> cell id per point -> stable radix sort of
//           (cell, point index) pairs (rocPRIM via hipCUB) -> points gathered into cell order as float4 (x, y, z,
//           original index) + cell_start[G^3 + 1] by binary search.
//   query:  one WAVE per 64 queries (an 8 x 8 pixel tile of a view, or 64 consecutive queries). The key of a candidate is
//           (d2, index), d2 = ((dx*dx + dy*dy) + dz*dz) in float32 without FMA - exactly the brute-force arithmetic - and
//           the 8 smallest keys win. A box of the grid (cell, 4^3 cells, 16^3 cells) or a point is skipped only when a lower
//           bound of its distance, shrunk by the safety factor 0.999 (far above float32 rounding), is STRICTLY above the
//           current 8th distance: nothing skipped can enter the result or tie with it, whatever the visiting order.
//           Coherent waves (queries within 6 cells of each other) search together, wave-uniformly (see the kernel);
//           scattered queries first walk Chebyshev shells R = 0..3 around their own cell, one lane per query, and stop
//           when the 8th distance lies inside the visited block; what is left joins the wave search.
// A rendered view is ~60 % background: those pixels' "3-D points" are near-plane points (NC:418-423: argmax of all-zero
// weights = sample 0) one to two scene units from every point of the set - tens of cells. The wave search ranks the blocks
// of 4^3 coarse cells (4^3 cells each) by the distance of their boxes and descends best first; ~1000 distances per query
// where the brute-force scan computes 1.92 M.
#include "common.h"

#include <hipcub/hipcub.hpp>

namespace nerfail {

struct Grid {
    float ox, oy, oz;     // origin (bbox min)
    float cs, inv_cs;     // cell size (cubic cells)
    int G;
};

__global__ __launch_bounds__(256) void bbox_kernel(const float* __restrict__ pts, long n, float* __restrict__ mm /*6: min xyz, max xyz*/) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = pts[3 * i + a];
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = wave_min(lo[a]);
        hi[a] = wave_max(hi[a]);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            // float atomic min / max through the ordered-integer trick
            if (lo[a] >= 0.f) atomicMin(reinterpret_cast<int*>(mm + a), __float_as_int(lo[a]));
            else atomicMax(reinterpret_cast<unsigned*>(mm + a), __float_as_uint(lo[a]));
            if (hi[a] >= 0.f) atomicMax(reinterpret_cast<int*>(mm + 3 + a), __float_as_int(hi[a]));
            else atomicMin(reinterpret_cast<unsigned*>(mm + 3 + a), __float_as_uint(hi[a]));
        }
    }
}

__global__ void bbox_init_kernel(float* mm) {
    if (threadIdx.x < 3) mm[threadIdx.x] = INFINITY;
    else if (threadIdx.x < 6) mm[threadIdx.x] = -INFINITY;
}

// grid parameters from the bbox (device side, so the build never synchronises with the host)
__global__ void grid_params_kernel(const float* __restrict__ mm, int G, Grid* __restrict__ g) {
    if (threadIdx.x != 0) return;
    float ext = fmaxf(fmaxf(mm[3] - mm[0], mm[4] - mm[1]), mm[5] - mm[2]);
    if (!(ext > 0.f)) ext = 1.f;
    const float cs = ext * 1.0001f / (float)G;
    g->ox = mm[0]; g->oy = mm[1]; g->oz = mm[2];
    g->cs = cs; g->inv_cs = 1.0f / cs; g->G = G;
}

__device__ __forceinline__ int cell_coord(float v, float o, float inv_cs, int G) {
    int c = (int)floorf((v - o) * inv_cs);
    return c < 0 ? 0 : (c >= G ? G - 1 : c);
}

__global__ __launch_bounds__(256) void cell_ids_kernel(const float* __restrict__ pts, long n, const Grid* __restrict__ gp,
                                                       unsigned* __restrict__ keys, int* __restrict__ vals) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Grid g = *gp;
    const int cx = cell_coord(pts[3 * i], g.ox, g.inv_cs, g.G), cy = cell_coord(pts[3 * i + 1], g.oy, g.inv_cs, g.G),
              cz = cell_coord(pts[3 * i + 2], g.oz, g.inv_cs, g.G);
    keys[i] = (unsigned)((cz * g.G + cy) * g.G + cx);
    vals[i] = (int)i;
}

__global__ __launch_bounds__(256) void gather_sorted_kernel(const float* __restrict__ pts, const int* __restrict__ order,
                                                            long n, float4* __restrict__ sorted) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int src = order[i];
    sorted[i] = make_float4(pts[3 * (long)src], pts[3 * (long)src + 1], pts[3 * (long)src + 2], __int_as_float(src));
}

__global__ __launch_bounds__(256) void cell_start_kernel(const unsigned* __restrict__ keys_sorted, long n, long ncells,
                                                         int* __restrict__ cell_start) {
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c > ncells) return;
    long lo = 0, hi = n;
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if ((long)keys_sorted[mid] < c) lo = mid + 1; else hi = mid;
    }
    cell_start[c] = (int)lo;
}

constexpr int kCoarse = 4;
constexpr int kSuper = 4;         // coarse cells per side of a super block
constexpr int kShellCap = 3;          // fine shells (beyond the query's own cell) before the coarse search takes over

// points per coarse cell = sum over its 16 x-rows of 4 consecutive fine cells
__global__ __launch_bounds__(256) void coarse_count_kernel(const int* __restrict__ cell_start, int G, int Gc, int* __restrict__ cnt) {
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (long)Gc * Gc * Gc) return;
    const int X = (int)(c % Gc), Y = (int)((c / Gc) % Gc), Z = (int)(c / ((long)Gc * Gc));
    const int x0 = X * kCoarse, x1 = min(x0 + kCoarse, G);
    int n = 0;
    for (int z = Z * kCoarse; z < min(Z * kCoarse + kCoarse, G); ++z)
        for (int y = Y * kCoarse; y < min(Y * kCoarse + kCoarse, G); ++y) {
            const long row = ((long)z * G + y) * G;
            n += cell_start[row + x1] - cell_start[row + x0];
        }
    cnt[c] = n;
}

// third level: points per block of kSuper^3 coarse cells (a few hundred blocks: the far-query search ranks ALL of them)
__global__ __launch_bounds__(256) void super_count_kernel(const int* __restrict__ coarse_cnt, int Gc, int Gs, int* __restrict__ cnt) {
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (long)Gs * Gs * Gs) return;
    const int X = (int)(c % Gs), Y = (int)((c / Gs) % Gs), Z = (int)(c / ((long)Gs * Gs));
    int n = 0;
    for (int z = Z * kSuper; z < min(Z * kSuper + kSuper, Gc); ++z)
        for (int y = Y * kSuper; y < min(Y * kSuper + kSuper, Gc); ++y)
            for (int x = X * kSuper; x < min(X * kSuper + kSuper, Gc); ++x) n += coarse_cnt[((long)z * Gc + y) * Gc + x];
    cnt[c] = n;
}

// A query's 8 best candidates are packed 64-bit keys, ascending: float32 bits of d2 (non-negative) in the high word, the point
// index in the low word - unsigned integer order == the lexicographic (d2, index) order of the brute-force kernel, one compare
// per test, and a sorted insert is 8 compare-exchange steps without branches.
typedef unsigned long long Key;
constexpr Key kWorstKey = ((Key)0x7f800000u << 32) | 0x7fffffffu;       // (d2 = +inf, index = int max)
__device__ __forceinline__ Key pack_key(float d2, int idx) { return ((Key)__float_as_uint(d2) << 32) | (unsigned)idx; }
__device__ __forceinline__ float key_d2(Key k) { return __uint_as_float((unsigned)(k >> 32)); }
__device__ __forceinline__ void insert_key(Key (&fk)[8], Key k) {       // fk stays sorted; the largest of the 9 keys drops out
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool c = k < fk[i];
        const Key lo = c ? k : fk[i];
        k = c ? fk[i] : k;
        fk[i] = lo;
    }
}

#ifndef NF_KNN_BATCH
#define NF_KNN_BATCH 8     // 1 (one load per candidate): 1.84 ms per 640 000-query view, 2: 1.61, 4: 1.37, 8: 1.27, 16: 1.33
#endif
// Candidates [b, e) of one contiguous cell range against the running top-8. NF_KNN_BATCH point loads are issued before the first
// is examined (clamped index, so the loads are unconditional): the search is latency bound, one dependent 16-byte load
// per candidate otherwise. The order in which candidates are examined does not matter: the key (d2, index) is total.
__device__ __forceinline__ void scan_points(const float4* __restrict__ sorted, int b, int e, float qx, float qy, float qz,
                                            Key (&fk)[8], unsigned& examined) {
    constexpr int U = NF_KNN_BATCH;
    examined += (unsigned)(e - b);
    for (int p = b; p < e; p += U) {
        float4 pt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) pt[u] = sorted[min(p + u, e - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (p + u < e) {
                const float dx = __fsub_rn(qx, pt[u].x), dy = __fsub_rn(qy, pt[u].y), dz = __fsub_rn(qz, pt[u].z);
                const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                const int id = __float_as_int(pt[u].w);
                const Key key = pack_key(d2, id);
                if (key < fk[7]) insert_key(fk, key);
            }
        }
    }
}

// wave-wide min / max on DPP moves (row_shr 1/2/4/8, row_bcast 15/31: an inclusive scan whose lane 63 holds the total),
// broadcast by v_readlane: the result is wave-uniform (an SGPR). __shfl_xor is ds_bpermute_b32 on gfx9 - six dependent trips
// through the LDS crossbar per reduction. NaN-free inputs.
template <int CTRL, int ROW_MASK, bool MAX>
__device__ __forceinline__ float dpp_minmax_step(float v) {
    const float o = __int_as_float(__builtin_amdgcn_update_dpp(MAX ? (int)0xff800000 : 0x7f800000, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
    return MAX ? fmaxf(v, o) : fminf(v, o);
}
template <bool MAX>
__device__ __forceinline__ float wave_minmax_uniform(float v) {
    v = dpp_minmax_step<0x111, 0xF, MAX>(v); v = dpp_minmax_step<0x112, 0xF, MAX>(v); v = dpp_minmax_step<0x114, 0xF, MAX>(v);
    v = dpp_minmax_step<0x118, 0xF, MAX>(v); v = dpp_minmax_step<0x142, 0xA, MAX>(v); v = dpp_minmax_step<0x143, 0xC, MAX>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

constexpr int kCoherentCells = 6;     // lanes whose queries span at most this many cells per axis search as ONE group (see below)
constexpr int kGroupPasses = 4;       // groups per wave before the rest goes lane by lane
constexpr int kMinGroup = 8;

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) void knn8_grid_kernel(const float* __restrict__ queries, long nq, int view_h, int view_w, const Grid* __restrict__ gp,
                                                        const float4* __restrict__ sorted, const int* __restrict__ cell_start,
                                                        const int* __restrict__ coarse_cnt, const int* __restrict__ super_cnt, float* __restrict__ dist, float* __restrict__ idx_f,
                                                        int* __restrict__ idx_i, unsigned long long* __restrict__ stats) {
    // One wave per workgroup. view_w > 0: the queries are the [view_h, view_w] pixels of a view and a wave takes an 8 x 8 pixel
    // TILE (0.014 scene units across on a lego view, a third of a cell: the 64 searches share nearly all their candidates);
    // view_w == 0: 64 consecutive queries. Lanes past the end repeat a valid query of the wave (no early return: the search
    // below is wave-cooperative) and store nothing.
    bool valid;
    long qi;
    if (view_w > 0) {
        const int tiles_x = (view_w + 7) >> 3;
        const int row = (int)(blockIdx.x / tiles_x) * 8 + (int)(threadIdx.x >> 3), col = (int)(blockIdx.x % tiles_x) * 8 + (int)(threadIdx.x & 7);
        valid = row < view_h && col < view_w;
        qi = (long)min(row, view_h - 1) * view_w + min(col, view_w - 1);
    } else {
        const long qi_raw = (long)blockIdx.x * 64 + threadIdx.x;
        valid = qi_raw < nq;
        qi = valid ? qi_raw : nq - 1;
    }
    const Grid g = *gp;
    const int G = g.G;
    const float qx = queries[3 * qi], qy = queries[3 * qi + 1], qz = queries[3 * qi + 2];
    const int cx = cell_coord(qx, g.ox, g.inv_cs, G), cy = cell_coord(qy, g.oy, g.inv_cs, G), cz = cell_coord(qz, g.oz, g.inv_cs, G);
    Key fk[8];                           // this lane's result (the shell walk and the wave search both work on it)
#pragma unroll
    for (int k = 0; k < 8; ++k) fk[k] = kWorstKey;
    unsigned examined = 0u;             // candidate points this query computed a distance to (reported through `stats`)

    const int Gc = (G + kCoarse - 1) / kCoarse, Gs = (Gc + kSuper - 1) / kSuper;
    // ---- scattered queries: one lane per query walks the Chebyshev shells R = 0..kShellCap around its own cell (returns
    // true when the query is finished: `fk` holds its result)
    auto shell_walk = [&]() -> bool {
        bool done = false;
        // a query more than kShellCap cells outside the grid cannot finish in the shell phase (its clamped cell's shells stay
        // farther than that from it): it goes straight to the ranked search (whatever the shells found is discarded there anyway)
        const float ext = g.cs * (float)G;
        const float outside = fmaxf(fmaxf(fmaxf(g.ox - qx, qx - (g.ox + ext)), fmaxf(g.oy - qy, qy - (g.oy + ext))),
                                    fmaxf(g.oz - qz, qz - (g.oz + ext)));
        bool skip_shells = outside > (float)(kShellCap + 1) * g.cs;
        if (!skip_shells) {
            // the 3 x 3 x 3 coarse cells around the query's coarse cell contain every fine cell of shells 0..kShellCap (kShellCap <
            // kCoarse): if they hold no point (a background pixel's near-plane point in empty space) the shell walk - ~100 dependent
            // lookups of empty cell ranges - is skipped. 27 independent loads.
            static_assert(kShellCap < kCoarse, "coarse neighbourhood must cover the shell block");
            const int CX = cx / kCoarse, CY = cy / kCoarse, CZ = cz / kCoarse;
            int near = 0;
#pragma unroll
            for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int X = min(max(CX + dx, 0), Gc - 1), Y = min(max(CY + dy, 0), Gc - 1), Z = min(max(CZ + dz, 0), Gc - 1);
                        near |= coarse_cnt[((long)Z * Gc + Y) * Gc + X];
                    }
            skip_shells = near == 0;
        }
        // a cell (or a whole x row of cells) whose box lies farther from the query than the current 8th key cannot change the
        // result (strict, same 0.999 safety factor as the stopping rule): it is not opened. After shell 0 the 8th distance of a
        // surface query is a fraction of the cell size, so of the 26 cells of shell 1 only the few the query is close to are
        // scanned - on view geometry (~100 points per surface cell) that is most of the candidates (round 3).
        auto cannot_improve_shell = [&](float d2box) {
            const float md = 0.999f * sqrtf(d2box);
            return key_d2(fk[7]) < md * md;
        };
        for (int R = 0; R < G && R <= kShellCap && !skip_shells; ++R) {
            const int z0 = max(cz - R, 0), z1 = min(cz + R, G - 1);
            const int y0 = max(cy - R, 0), y1 = min(cy + R, G - 1);
            const int x0 = max(cx - R, 0), x1 = min(cx + R, G - 1);
            for (int z = z0; z <= z1; ++z) {
                const bool zface = (z == cz - R) || (z == cz + R);
                const float cz0 = g.oz + (float)z * g.cs;
                const float ez = fmaxf(fmaxf(cz0 - qz, qz - (cz0 + g.cs)), 0.f);
                for (int y = y0; y <= y1; ++y) {
                    const bool yface = zface || (y == cy - R) || (y == cy + R);
                    const float cy0 = g.oy + (float)y * g.cs;
                    const float ey = fmaxf(fmaxf(cy0 - qy, qy - (cy0 + g.cs)), 0.f);
                    const float eyz = ey * ey + ez * ez;
                    if (R > 0 && cannot_improve_shell(eyz)) continue;              // the whole row is too far
                    // on a z / y face of the shell the whole x row belongs to it; otherwise only the two end cells x = cx -+ R do
                    const long row = ((long)z * G + y) * G;
                    if (yface) {
                        if (R == 0) {
                            scan_points(sorted, cell_start[row + x0], cell_start[row + x1 + 1], qx, qy, qz, fk, examined);
                        } else {
                            int b = cell_start[row + x0];
                            for (int x = x0; x <= x1; ++x) {
                                const int e = cell_start[row + x + 1];
                                const float cx0 = g.ox + (float)x * g.cs;
                                const float ex = fmaxf(fmaxf(cx0 - qx, qx - (cx0 + g.cs)), 0.f);
                                if (e > b && !cannot_improve_shell(ex * ex + eyz)) scan_points(sorted, b, e, qx, qy, qz, fk, examined);
                                b = e;
                            }
                        }
                    } else {
#pragma unroll
                        for (int side = 0; side < 2; ++side) {
                            const int x = side ? cx + R : cx - R;
                            if (x < 0 || x >= G || (side == 1 && R == 0)) continue;
                            const float cx0 = g.ox + (float)x * g.cs;
                            const float ex = fmaxf(fmaxf(cx0 - qx, qx - (cx0 + g.cs)), 0.f);
                            if (cannot_improve_shell(ex * ex + eyz)) continue;
                            scan_points(sorted, cell_start[row + x], cell_start[row + x + 1], qx, qy, qz, fk, examined);
                        }
                    }
                }
            }
            // every unvisited point is outside the block [c-R, c+R]^3: lower bound of its distance to the query
            float bound = INFINITY;
            if (cx - R > 0) bound = fminf(bound, qx - (g.ox + (float)(cx - R) * g.cs));
            if (cx + R < G - 1) bound = fminf(bound, (g.ox + (float)(cx + R + 1) * g.cs) - qx);
            if (cy - R > 0) bound = fminf(bound, qy - (g.oy + (float)(cy - R) * g.cs));
            if (cy + R < G - 1) bound = fminf(bound, (g.oy + (float)(cy + R + 1) * g.cs) - qy);
            if (cz - R > 0) bound = fminf(bound, qz - (g.oz + (float)(cz - R) * g.cs));
            if (cz + R < G - 1) bound = fminf(bound, (g.oz + (float)(cz + R + 1) * g.cs) - qz);
            if (bound == INFINITY) { done = true; break; }      // the block covers the whole grid
            if (bound > 0.f) {
                const float sb = 0.999f * bound;
                if (key_d2(fk[7]) < sb * sb) { done = true; break; } // strict: nothing unvisited can enter or tie
            }
        }
        return done;
    };
    // ---- far queries (background pixels' points on the near plane: one to two units from every point, tens of cells): no shell
    // walk - the blocks of kSuper^3 coarse cells are RANKED by the distance of their boxes, a block / coarse cell / fine cell
    // is opened only if its box can still hold a point nearer than the lane's current 8th (strict, same 0.999 safety factor
    // as the stopping rule). Every block is either opened or excluded by a valid lower bound, so the result is exact whatever
    // the visiting order (the key (d2, index) is total).
    // Round 3: the ranking is done by the WAVE for its 64 queries together (they are neighbouring pixels: the same few blocks
    // matter to all of them). Round 2 let every lane walk all Gs^3 blocks twice on its own: 1000 dependent count loads per
    // query, 8 ms per view for the 380 000 background queries. Now each lane bounds Gs^3 / 64 blocks against the BOX of the
    // wave's far queries (a lower bound for each of them), the nearest non-empty block seeds every lane's top-8, and only the
    // blocks that can still matter to ANY lane (ballot) are walked - by all lanes together, so the count and cell-range
    // loads are wave-uniform - with the per-lane box tests deciding what each lane actually scans.
    // ---- the wave search: the lanes of `far` are searched TOGETHER (they lie within a few cells of each other, or are what the
    // shell walk left over)
    auto wave_search = [&](const bool far) {
        if (far) {                                                           // (whatever a shell walk found for these lanes is found again)
#pragma unroll
            for (int k = 0; k < 8; ++k) fk[k] = kWorstKey;
        }
        auto d8 = [&]() { return key_d2(fk[7]); };
        constexpr int kPark = 4;
        Key pend[kPark];                                                     // candidates waiting to be inserted (below)
        int n_pend = 0;
        // The WHOLE walk is wave-uniform. Blocks, coarse cells, fine cells and candidate points are visited by the wave in one
        // common order; a lane only contributes its own box tests - "can this box still hold a point nearer than MY 8th?" - and
        // a box is opened when ANY lane of the group says yes (ballot). Every lane of the group then measures every candidate
        // the wave lets through: extra candidates never change an exact result, the lanes' bounds tighten together, and 64
        // neighbouring pixels' searches cost one instruction stream instead of 64 divergent chains of dependent loads.
        const float ccs = g.cs * (float)kCoarse, scs = ccs * (float)kSuper;
        auto any_needs = [&](float d2box) {                                  // does any far lane still need a box at this distance?
            const float md = 0.999f * sqrtf(d2box);
            return __ballot(far && !(d8() < md * md)) != 0ull;
        };
        auto axis_d = [](float lo, float size, float q) { return fmaxf(fmaxf(lo - q, q - (lo + size)), 0.f); };
        // box of the wave's far queries
        const float blx = wave_minmax_uniform<false>(far ? qx : INFINITY), bhx = wave_minmax_uniform<true>(far ? qx : -INFINITY);
        const float bly = wave_minmax_uniform<false>(far ? qy : INFINITY), bhy = wave_minmax_uniform<true>(far ? qy : -INFINITY);
        const float blz = wave_minmax_uniform<false>(far ? qz : INFINITY), bhz = wave_minmax_uniform<true>(far ? qz : -INFINITY);
        const int lane = threadIdx.x & 63;
        auto lane_i = [](int v, int l) { return __builtin_amdgcn_readlane(v, l); };
        auto lane_f = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
        auto wave_box_d2 = [&](float x0, float y0, float z0, float size) {   // lower bound of |q - p|, p in the cube, for EVERY far query of the wave
            const float ddx = fmaxf(fmaxf(x0 - bhx, blx - (x0 + size)), 0.f);
            const float ddy = fmaxf(fmaxf(y0 - bhy, bly - (y0 + size)), 0.f);
            const float ddz = fmaxf(fmaxf(z0 - bhz, blz - (z0 + size)), 0.f);
            return ddx * ddx + ddy * ddy + ddz * ddz;
        };
        auto wave_d8max = [&]() { return wave_minmax_uniform<true>(far ? d8() : 0.f); };
        // A candidate that passes a lane's threshold is only PARKED in one of the lane's kPark slots; the sorted inserts run for
        // the whole wave at once when some lane needs a fifth slot (and at the end of the cell). With 64 lanes almost every
        // candidate of the first cells improves SOME lane, and the wave paid a masked 8-step insert nearly every iteration
        // (a surface tile: ~290 insert rounds for ~1000 candidates); parked, it inserts a few dozen times per
        // search. Parked keys leave the lane's threshold stale (too large) until they are inserted: more candidates pass,
        // none is lost.
        auto flush = [&]() {
#pragma unroll
            for (int sl = 0; sl < kPark; ++sl) {
                if (__ballot(sl < n_pend) == 0ull) break;
                if (sl < n_pend) insert_key(fk, pend[sl]);
            }
            n_pend = 0;
        };
        auto park = [&](bool p, Key key) {                                   // p: this lane keeps `key`
            if (__ballot(p & (n_pend == kPark)) != 0ull) {
                flush();
                p = p & (key < fk[7]);
            }
            if (p) {
#pragma unroll
                for (int sl = 0; sl < kPark; ++sl)
                    if (n_pend == sl) pend[sl] = key;
                n_pend += 1;
            }
        };
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
        // Candidates [b, e) (uniform) against every far lane's top-8. The 64 lanes fetch 64 consecutive points with ONE load
        // (1 KB, coalesced) and each lane first bounds ITS point against the box of the wave's queries - a lower bound of its
        // distance to every one of them, a fraction of a cell away from the true distances when the wave is a pixel tile:
        // points that cannot reach the largest 8th distance of the wave drop out here, 64 per instruction. The survivors
        // (ballot) are broadcast from their lanes' registers (v_readlane), four per trip as two float pairs (v_pk_*
        // arithmetic, no contraction), filtered by one float compare against the lane's 8th distance (<=: ties are settled
        // on the packed key); the point index is only fetched for candidates that pass.
        auto scan_uniform = [&](int b, int e) {
            for (int p0 = b; p0 < e; p0 += 64) {
                const int n = min(64, e - p0);
                const float4 mine = sorted[min(p0 + lane, e - 1)];
                const float bx = fmaxf(fmaxf(blx - mine.x, mine.x - bhx), 0.f), by = fmaxf(fmaxf(bly - mine.y, mine.y - bhy), 0.f),
                            bz = fmaxf(fmaxf(blz - mine.z, mine.z - bhz), 0.f);
                const float lb = 0.999f * sqrtf(bx * bx + by * by + bz * bz);
                const float d8max = wave_d8max();                             // (all lanes: a cross-lane reduction)
                unsigned long long m = __ballot(lane < n && !(d8max < lb * lb));
                if (far) examined += (unsigned)__popcll(m);
                while (m != 0ull) {
                    int l[4];
                    bool ok[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        ok[u] = m != 0ull;
                        l[u] = ok[u] ? __ffsll((long long)m) - 1 : l[0];
                        m &= m - 1ull;                                        // (0 stays 0)
                    }
                    const float d8f = d8();
                    f32x2 d2p[2];
                    bool any = false;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int l0 = l[2 * h], l1 = l[2 * h + 1];
                        const f32x2 px = {lane_f(mine.x, l0), lane_f(mine.x, l1)}, py = {lane_f(mine.y, l0), lane_f(mine.y, l1)},
                                    pz = {lane_f(mine.z, l0), lane_f(mine.z, l1)};
                        const f32x2 dx = qx2 - px, dy = qy2 - py, dz = qz2 - pz;
                        d2p[h] = (dx * dx + dy * dy) + dz * dz;
                        any |= (d2p[h].x <= d8f) | (d2p[h].y <= d8f);
                    }
                    if (__ballot(far & any) == 0ull) continue;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (!ok[u]) break;
                        const float d2u = u & 1 ? d2p[u >> 1].y : d2p[u >> 1].x;
                        if (__ballot(far & (d2u <= d8f)) == 0ull) continue;       // (usually one of the four is the one that passed)
                        const unsigned id = (unsigned)lane_i(__float_as_int(mine.w), l[u]);
                        const Key key = ((Key)__float_as_uint(d2u) << 32) | id;
                        park(far & (key < fk[7]), key);
                    }
                }
            }
            if (__ballot(n_pend != 0) != 0ull) flush();
        };
        // A 4 x 4 x 4 group of boxes is looked at by the 64 lanes together, lane = box: its count / point range arrives with
        // one or two loads per WAVE (one round trip, where a loop over the boxes waited for 64 dependent loads), its distance
        // to the box of the wave's queries is a lower bound for every lane. Boxes are opened BEST FIRST: the nearest one that
        // can still matter to the lane with the largest 8th distance, then the bounds are looked at again - after the first
        // non-empty box of a surface wave the 8th distances are a fraction of a cell and nearly everything else drops out in
        // that parallel test; what survives it is tested against every lane's own query before it is opened.
        auto best_first = [&](bool cand, float bd2, auto&& open) {
            const float md = 0.999f * sqrtf(bd2), mdsq = md * md;
            while (true) {
                const float d8max = wave_d8max();
                const bool alive = cand && !(d8max < mdsq);
                if (__ballot(alive) == 0ull) break;
                const float v = alive ? bd2 : INFINITY;
                const float mn = wave_minmax_uniform<false>(v);
                const int pick = __ffsll((long long)__ballot(alive && v == mn)) - 1;
                cand = cand && lane != pick;
                open(pick);
            }
        };
        auto visit_coarse = [&](int X, int Y, int Z) {                        // lane = fine cell of the coarse cell
            const int fx0 = X * kCoarse, fy0 = Y * kCoarse, fz0 = Z * kCoarse;
            const int x = fx0 + (lane & 3), y = fy0 + ((lane >> 2) & 3), z = fz0 + (lane >> 4);
            int cb = 0, ce = 0;
            if (x < G && y < G && z < G) {
                const long at = ((long)z * G + y) * G + x;
                cb = cell_start[at]; ce = cell_start[at + 1];
            }
            best_first(ce > cb, wave_box_d2(g.ox + (float)x * g.cs, g.oy + (float)y * g.cs, g.oz + (float)z * g.cs, g.cs), [&](int pick) {
                const float ex = axis_d(g.ox + (float)(fx0 + (pick & 3)) * g.cs, g.cs, qx), ey = axis_d(g.oy + (float)(fy0 + ((pick >> 2) & 3)) * g.cs, g.cs, qy),
                            ez = axis_d(g.oz + (float)(fz0 + (pick >> 4)) * g.cs, g.cs, qz);
                if (any_needs(ex * ex + ey * ey + ez * ez)) scan_uniform(lane_i(cb, pick), lane_i(ce, pick));
            });
        };
        auto box_d2 = [&](float x0, float y0, float z0, float size) {      // squared distance from the lane's query to an axis-aligned cube
            const float ddx = axis_d(x0, size, qx), ddy = axis_d(y0, size, qy), ddz = axis_d(z0, size, qz);
            return ddx * ddx + ddy * ddy + ddz * ddz;
        };
        auto visit_super = [&](int S) {                                       // lane = coarse cell of the block
            const int SX = S % Gs, SY = (S / Gs) % Gs, SZ = S / (Gs * Gs);
            if (!any_needs(box_d2(g.ox + (float)SX * scs, g.oy + (float)SY * scs, g.oz + (float)SZ * scs, scs))) return;
            const int X0 = SX * kSuper, Y0 = SY * kSuper, Z0 = SZ * kSuper;
            const int Xc = X0 + (lane & 3), Yc = Y0 + ((lane >> 2) & 3), Zc = Z0 + (lane >> 4);
            const bool cand = Xc < Gc && Yc < Gc && Zc < Gc && coarse_cnt[((long)Zc * Gc + Yc) * Gc + Xc] != 0;
            best_first(cand, wave_box_d2(g.ox + (float)Xc * ccs, g.oy + (float)Yc * ccs, g.oz + (float)Zc * ccs, ccs), [&](int pick) {
                const int Xu = X0 + (pick & 3), Yu = Y0 + ((pick >> 2) & 3), Zu = Z0 + (pick >> 4);
                if (any_needs(box_d2(g.ox + (float)Xu * ccs, g.oy + (float)Yu * ccs, g.oz + (float)Zu * ccs, ccs))) visit_coarse(Xu, Yu, Zu);
            });
        };
        auto box_box_d2 = [&](int S) {
            const int SX = S % Gs, SY = (S / Gs) % Gs, SZ = S / (Gs * Gs);
            return wave_box_d2(g.ox + (float)SX * scs, g.oy + (float)SY * scs, g.oz + (float)SZ * scs, scs);
        };
        const int nS = Gs * Gs * Gs;
        // 1. the nearest non-empty block (by the wave's box) seeds the top-8 of every far lane (each lane ranks nS / 64 blocks)
        float best_d2 = INFINITY;
        int best = 0x7fffffff;
        for (int S = lane; S < nS; S += 64) {
            if (super_cnt[S] == 0) continue;
            const float d2 = box_box_d2(S);
            if (d2 < best_d2 || (d2 == best_d2 && S < best)) { best_d2 = d2; best = S; }
        }
        const float wmin = wave_minmax_uniform<false>(best_d2);
        const unsigned long long at = __ballot(best_d2 == wmin && best != 0x7fffffff);
        int seed = -1;
        if (at != 0ull) seed = __builtin_amdgcn_readfirstlane(__shfl(best, __ffsll((long long)at) - 1, 64));
        if (seed >= 0) visit_super(seed);
        // 2. every other block that can still hold a point nearer than the worst lane's 8th
        for (int base = 0; base < nS; base += 64) {
            const float d8max = wave_d8max();                                 // (refreshed per batch: the lanes' bounds only shrink)
            const int S = base + lane;
            bool cand = S < nS && S != seed && super_cnt[S] != 0;
            if (cand) {
                const float md = 0.999f * sqrtf(box_box_d2(S));
                cand = !(d8max < md * md);
            }
            unsigned long long m = __ballot(cand);
            while (m != 0ull) {
                const int bit = __ffsll((long long)m) - 1;
                m &= m - 1ull;
                visit_super(base + bit);
            }
        }
    };
    // The 64 queries of a wave are neighbouring pixels of a view. Up to kGroupPasses times the first unsolved lane names a
    // group - the unsolved lanes within kCoherentCells / 2 cells of its query - and the wave searches for the whole group at
    // once: one instruction stream, coalesced loads, shared candidates. A pixel tile is one group; a tile across a silhouette
    // is two or three (surface / background). Groups of fewer than kMinGroup lanes are not worth a wave-wide search: what is
    // left then takes the per-lane shell walk, and the wave search once more for the queries that did not finish there.
    bool pending = true;
    unsigned long long searched = 0ull;
    for (int pass = 0; __ballot(pending) != 0ull; ++pass) {
        bool grp;
        if (pass < kGroupPasses) {
            const int sl = __ffsll((long long)__ballot(pending)) - 1;
            const float sx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qx), sl)), sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qy), sl)),
                        sz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qz), sl));
            grp = pending && fmaxf(fmaxf(fabsf(qx - sx), fabsf(qy - sy)), fabsf(qz - sz)) <= 0.5f * (float)kCoherentCells * g.cs;
            if (__popcll(__ballot(grp)) < kMinGroup) { pass = kGroupPasses - 1; continue; }
        } else {
            bool done = false;
            if (pending) done = shell_walk();
            grp = pending && !done;
            pending = grp;
        }
        searched += __popcll(__ballot(valid && grp));
        if (__ballot(grp) != 0ull) wave_search(grp);
        pending = pending && !grp;
    }
    if (stats != nullptr) {             // [0] candidates examined, [1] queries that took the far search (one atomic per wave)
        unsigned long long tot = valid ? examined : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
        if ((threadIdx.x & 63) == 0) { atomicAdd(stats, tot); atomicAdd(stats + 1, searched); }
    }
    if (!valid) return;
    // 32 bytes per query and output as two 16-byte stores (eight 4-byte stores 32 bytes apart wrote every 64-byte line of the
    // outputs in pieces: 5x the bytes at the memory side)
    float d[8], jf[8];
    int ji[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        d[k] = sqrt_rn(key_d2(fk[k]));
        ji[k] = (int)(unsigned)fk[k];
        jf[k] = (float)ji[k];
    }
    float4* __restrict__ od = reinterpret_cast<float4*>(dist + 8 * qi);
    od[0] = make_float4(d[0], d[1], d[2], d[3]);
    od[1] = make_float4(d[4], d[5], d[6], d[7]);
    if (idx_f != nullptr) {
        float4* __restrict__ of = reinterpret_cast<float4*>(idx_f + 8 * qi);
        of[0] = make_float4(jf[0], jf[1], jf[2], jf[3]);
        of[1] = make_float4(jf[4], jf[5], jf[6], jf[7]);
    }
    if (idx_i != nullptr) {
        int4* __restrict__ oi = reinterpret_cast<int4*>(idx_i + 8 * qi);
        oi[0] = make_int4(ji[0], ji[1], ji[2], ji[3]);
        oi[1] = make_int4(ji[4], ji[5], ji[6], ji[7]);
    }
}

static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

static int grid_dim_for(long n) {
    int G = (int)lround(cbrt((double)n));
    if (G < 4) G = 4;
    if (G > 256) G = 256;
    return G;
}

static size_t coarse_cells(int G) { const size_t Gc = (size_t)(G + kCoarse - 1) / kCoarse; return Gc * Gc * Gc; }
static size_t super_cells(int G) { const size_t Gc = (size_t)(G + kCoarse - 1) / kCoarse, Gs = (Gc + kSuper - 1) / kSuper; return Gs * Gs * Gs; }

static size_t knn_cub_temp(long n) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr,
                                       (int*)nullptr, (int)n, 0, 32, (hipStream_t) nullptr);
    return bytes;
}

}  // namespace nerfail

using namespace nerfail;

extern "C" size_t nerfail_knn8_grid_workspace_bytes(int64_t n_points) {
    if (n_points < NERFAIL_KNN || n_points >= (1 << 24)) return 0;
    const int G = grid_dim_for(n_points);
    const size_t ncells = (size_t)G * G * G;
    return al256(256) /* bbox + Grid */ + 4 * al256((size_t)n_points * 4) /* keys in/out, vals in/out */ +
           al256((size_t)n_points * 16) /* sorted float4 */ + al256((ncells + 1) * 4) + al256(coarse_cells(G) * 4) +
           al256(super_cells(G) * 4) + al256(knn_cub_temp(n_points));
}

// The grid of a point set: built once (the set of a scene is fixed - CI:57-61 stacks the base views once, CI:110-163 then
// walks 400 views against it), searched per view.
struct GridWs {
    float* mm; Grid* gp;
    unsigned *keys_in, *keys_out; int *vals_in, *vals_out;
    float4* sorted; int *cell_start, *coarse_cnt, *super_cnt;
    void* temp; size_t temp_bytes;
    int G;
};
static GridWs carve(void* workspace, long n) {
    GridWs w;
    const int G = grid_dim_for(n);
    const long ncells = (long)G * G * G;
    char* ws = (char*)workspace;
    w.G = G;
    w.mm = (float*)ws;
    w.gp = (Grid*)(ws + 64);
    ws += al256(256);
    const size_t seg = al256((size_t)n * 4);
    w.keys_in = (unsigned*)ws; w.keys_out = (unsigned*)(ws + seg);
    w.vals_in = (int*)(ws + 2 * seg); w.vals_out = (int*)(ws + 3 * seg);
    ws += 4 * seg;
    w.sorted = (float4*)ws; ws += al256((size_t)n * 16);
    w.cell_start = (int*)ws; ws += al256((size_t)(ncells + 1) * 4);
    w.coarse_cnt = (int*)ws; ws += al256(coarse_cells(G) * 4);
    w.super_cnt = (int*)ws; ws += al256(super_cells(G) * 4);
    w.temp = ws;
    w.temp_bytes = knn_cub_temp(n);
    return w;
}

extern "C" int nerfail_knn8_grid_build(const float* points, int64_t n_points, void* workspace, size_t workspace_bytes, void* stream) {
    NF_REQUIRE(n_points >= NERFAIL_KNN, "need at least 8 points");
    NF_REQUIRE(n_points < (1 << 24), "n_points must be < 2^24 (indices are stored as float32, CI:148-163)");
    NF_REQUIRE(points != nullptr, "NULL pointer");
    NF_REQUIRE(workspace != nullptr && workspace_bytes >= nerfail_knn8_grid_workspace_bytes(n_points),
               "workspace too small (nerfail_knn8_grid_workspace_bytes)");
    hipStream_t s = as_stream(stream);
    const long n = n_points;
    const GridWs w = carve(workspace, n);
    const int G = w.G;
    const long ncells = (long)G * G * G;
    bbox_init_kernel<<<dim3(1), dim3(64), 0, s>>>(w.mm);
    NF_LAUNCHED("bbox_init_kernel");
    bbox_kernel<<<dim3(256), dim3(256), 0, s>>>(points, n, w.mm);
    NF_LAUNCHED("bbox_kernel");
    grid_params_kernel<<<dim3(1), dim3(64), 0, s>>>(w.mm, G, w.gp);
    NF_LAUNCHED("grid_params_kernel");
    cell_ids_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(points, n, w.gp, w.keys_in, w.vals_in);
    NF_LAUNCHED("cell_ids_kernel");
    int bits = 1;
    while ((1L << bits) < ncells) ++bits;
    size_t temp_bytes = w.temp_bytes;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(w.temp, temp_bytes, w.keys_in, w.keys_out, w.vals_in, w.vals_out, (int)n, 0, bits, s);
    if (e != hipSuccess) return hip_fail(e, "hipcub::DeviceRadixSort::SortPairs");
    gather_sorted_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(points, w.vals_out, n, w.sorted);
    NF_LAUNCHED("gather_sorted_kernel");
    cell_start_kernel<<<dim3((unsigned)((ncells + 1 + 255) / 256)), dim3(256), 0, s>>>(w.keys_out, n, ncells, w.cell_start);
    NF_LAUNCHED("cell_start_kernel");
    const int Gc = (G + kCoarse - 1) / kCoarse;
    coarse_count_kernel<<<dim3((unsigned)((coarse_cells(G) + 255) / 256)), dim3(256), 0, s>>>(w.cell_start, G, Gc, w.coarse_cnt);
    NF_LAUNCHED("coarse_count_kernel");
    const int Gs = (Gc + kSuper - 1) / kSuper;
    super_count_kernel<<<dim3((unsigned)((super_cells(G) + 255) / 256)), dim3(256), 0, s>>>(w.coarse_cnt, Gc, Gs, w.super_cnt);
    NF_LAUNCHED("super_count_kernel");
    return NERFAIL_OK;
}

// optional work counters of the search (NULL = off): set with nerfail_knn8_grid_stats; two device uint64 that the caller zeroes
static unsigned long long* g_knn_stats = nullptr;
extern "C" int nerfail_knn8_grid_stats(unsigned long long* stats) { g_knn_stats = stats; return NERFAIL_OK; }

static int grid_search(const float* queries, int64_t n_queries, int view_h, int view_w, int64_t n_points, float* dist, float* idx_f32,
                       int32_t* idx_i32, const void* workspace, size_t workspace_bytes, void* stream) {
    NF_REQUIRE(n_queries >= 0, "n_queries is negative");
    NF_REQUIRE(n_points >= NERFAIL_KNN && n_points < (1 << 24), "bad n_points");
    if (n_queries == 0) return NERFAIL_OK;
    NF_REQUIRE(queries != nullptr && dist != nullptr, "NULL pointer");
    NF_REQUIRE(idx_f32 != nullptr || idx_i32 != nullptr, "need idx_f32 or idx_i32");
    NF_REQUIRE(((reinterpret_cast<uintptr_t>(dist) | reinterpret_cast<uintptr_t>(idx_f32) | reinterpret_cast<uintptr_t>(idx_i32)) & 15) == 0,
               "dist / idx must be 16-byte aligned");
    NF_REQUIRE(workspace != nullptr && workspace_bytes >= nerfail_knn8_grid_workspace_bytes(n_points),
               "workspace too small (nerfail_knn8_grid_workspace_bytes)");
    const GridWs w = carve(const_cast<void*>(workspace), n_points);
    const long waves = view_w > 0 ? (long)((view_w + 7) / 8) * ((view_h + 7) / 8) : (n_queries + 63) / 64;
    NF_REQUIRE(waves < (1L << 31), "too many queries for one launch");
    knn8_grid_kernel<<<dim3((unsigned)waves), dim3(64), 0, as_stream(stream)>>>(
        queries, n_queries, view_h, view_w, w.gp, w.sorted, w.cell_start, w.coarse_cnt, w.super_cnt, dist, idx_f32, idx_i32,
        g_knn_stats);
    NF_LAUNCHED("knn8_grid_kernel");
    return NERFAIL_OK;
}

extern "C" int nerfail_knn8_grid_search(const float* queries, int64_t n_queries, int64_t n_points, float* dist, float* idx_f32,
                                        int32_t* idx_i32, const void* workspace, size_t workspace_bytes, void* stream) {
    return grid_search(queries, n_queries, 0, 0, n_points, dist, idx_f32, idx_i32, workspace, workspace_bytes, stream);
}

extern "C" int nerfail_knn8_grid_search_view(const float* queries, int height, int width, int64_t n_points, float* dist,
                                             float* idx_f32, int32_t* idx_i32, const void* workspace, size_t workspace_bytes,
                                             void* stream) {
    NF_REQUIRE(height > 0 && width > 0, "bad view size");
    return grid_search(queries, (int64_t)height * width, height, width, n_points, dist, idx_f32, idx_i32, workspace, workspace_bytes, stream);
}

extern "C" int nerfail_knn8_grid(const float* queries, int64_t n_queries, const float* points, int64_t n_points, float* dist,
                                 float* idx_f32, int32_t* idx_i32, void* workspace, size_t workspace_bytes, void* stream) {
    NF_REQUIRE(n_queries >= 0, "n_queries is negative");
    NF_REQUIRE(n_points >= NERFAIL_KNN, "need at least 8 points");
    NF_REQUIRE(n_points < (1 << 24), "n_points must be < 2^24 (indices are stored as float32, CI:148-163)");
    if (n_queries == 0) return NERFAIL_OK;
    NF_REQUIRE(queries != nullptr && points != nullptr && dist != nullptr, "NULL pointer");
    NF_REQUIRE(idx_f32 != nullptr || idx_i32 != nullptr, "need idx_f32 or idx_i32");
    const int rc = nerfail_knn8_grid_build(points, n_points, workspace, workspace_bytes, stream);
    if (rc != NERFAIL_OK) return rc;
    return nerfail_knn8_grid_search(queries, n_queries, n_points, dist, idx_f32, idx_i32, workspace, workspace_bytes, stream);
}
