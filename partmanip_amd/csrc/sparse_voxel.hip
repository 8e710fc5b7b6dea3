// Sparse-voxel U-Net building blocks (`network.name: SparseUNet`, the "3D Sparse-UNet" backbone the reference's README
// names -- README.md:30 -- whose code is NOT in the snapshot, README.md:23: parity unpinned, checked against the test-side CPU
// restatement).  The observation is the reference's 'depth_sparse' format (tasks/hand_base.py:335-336,
// utils/depth2tsdf.py:88-120): P rows (x, y, z, f) per env with integer voxel coordinates in [0, R).
//
// MI355X-shaped design: geometry and arithmetic are separated.
//   * Geometry (integer work, HBM-bound, once per forward): a DENSE index grid per cloud and level (R^3 int32: 0.5 MB
//     per cloud at R = 50 -- 1 GB for a 2048-cloud mini-batch is small change next to 288 GB of HBM and makes every
//     neighbour / parent / child lookup one load instead of a hash probe), the 27-neighbour table of a submanifold
//     convolution, and the 2x2x2 parent / child tables of the strided levels.  Duplicate coordinates (the padding rows of
//     `sparse_voxel` all read voxel (0,0,0)) resolve to the LOWEST row index (atomicMin), everything is deterministic.
//   * Arithmetic: a sparse convolution is "gather the neighbour rows side by side" (pm_rows_gather_f32: a row of
//     J x C floats per output voxel, zeros for absent neighbours) followed by the fp32-MFMA Linear kernels of
//     gemm2_f32.hip on (rows x J C) x (J C x C_out) -- one big GEMM per layer instead of J small ones, bias + tanh in its
//     epilogue; the backward is the Linear backward kernels plus pm_rows_gather_bwd_f32, which routes the column
//     blocks back through the MIRRORED table as a gather (no atomics: fixed summation order), folds tanh' and can
//     accumulate a second contribution (skip connections).
#include "common.h"

#define VX_EMPTY 0x7fffffff

__global__ __launch_bounds__(256) void vx_fill_kernel(int32_t* __restrict__ p, long n, int32_t v) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = v;
}
static inline int vx_blocks(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// level 0: rows -> coords (b, x, y, z) and grid[b][cell] = lowest row with that coordinate
__global__ __launch_bounds__(256) void vx_grid0_kernel(const float* __restrict__ x, long ldx, int P, int C, int R, long rows,
                                                        int32_t* __restrict__ grid, int32_t* __restrict__ coords) {
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
        const long b = r / P;
        const float* q = x + b * ldx + (r - b * P) * C;
        int c[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            int v = (int)floorf(q[d]);
            c[d] = v < 0 ? 0 : (v >= R ? R - 1 : v);
        }
        coords[r * 4] = (int)b; coords[r * 4 + 1] = c[0]; coords[r * 4 + 2] = c[1]; coords[r * 4 + 3] = c[2];
        atomicMin(&grid[b * R * R * R + ((long)c[0] * R + c[1]) * R + c[2]], (int)r);
    }
}

// 27 neighbours of every row: nbr[r][o], o = (dx+1)*9 + (dy+1)*3 + (dz+1); offset 26 - o is the mirrored one.  ld >= 27 is the
// table's row stride: columns 27 .. ld-1 are written as -1 ("absent" taps that pad the gathered operand of a layer whose
// 27 * C_in is not a whole number of K-steps -- conv0: 27 x 4 = 108 -> 32 taps = 128 columns).
// (Measured SLOWER than this one-entry-per-thread loop: four entries per thread and trip with their loads issued back to back
// (-14 %); one thread per (row, dx, dy) covering the three dz neighbours -- a third of the index arithmetic, three consecutive
// stores per thread -- cost the SparseUNet forward +0.5 ms: the coalesced 4-byte store stream is what this kernel lives on.)
__global__ __launch_bounds__(256) void vx_nbr27_kernel(const int32_t* __restrict__ coords, long rows, const int32_t* __restrict__ grid,
                                                        int R, int32_t* __restrict__ nbr, int ld) {
    // (XCD-contiguous block order: a cloud's grid lines are then looked up through ONE XCD's L2 -- 2.40 -> 2.11 GB per launch at 2048 clouds)
    for (long e = (long)pm_xcd_contiguous(blockIdx.x, gridDim.x) * 256 + threadIdx.x; e < rows * ld; e += (long)gridDim.x * 256) {
        const long r = e / ld;
        const int o = (int)(e - r * ld);
        int v = -1;
        if (o < 27) {
            const int dx = o / 9 - 1, dy = (o / 3) % 3 - 1, dz = o % 3 - 1;
            const int b = coords[r * 4], X = coords[r * 4 + 1] + dx, Y = coords[r * 4 + 2] + dy, Z = coords[r * 4 + 3] + dz;
            if (X >= 0 && X < R && Y >= 0 && Y < R && Z >= 0 && Z < R) {
                const int g = grid[(long)b * R * R * R + ((long)X * R + Y) * R + Z];
                v = g == VX_EMPTY ? -1 : g;
            }
        }
        nbr[e] = v;
    }
}
// The table through which the DATA gradient of a 3^3 submanifold convolution gathers: row r reads s as neighbour o  <=>  s
// reads r as neighbour 26 - o, so the rows whose output used r are nbr[r] reversed.  A row that is nobody's neighbour (a
// duplicate coordinate: its own-cell entry points at the canonical row, not at itself) gets -1 everywhere.
__global__ __launch_bounds__(256) void vx_mirror27_kernel(const int32_t* __restrict__ nbr, long rows, int32_t* __restrict__ out) {
    const long total = rows * 27;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / 27;
        const int o = (int)(e - r * 27);
        const bool canon = nbr[r * 27 + 13] == (int)r;
        out[e] = canon ? nbr[r * 27 + 26 - o] : -1;
    }
}

// strided level: mark the parent cells of the fine rows, then count them per cloud (cell order)
__global__ __launch_bounds__(256) void vx_mark_kernel(const int32_t* __restrict__ coords, long rows, int Rc, int32_t* __restrict__ gridc) {
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
        const int b = coords[r * 4], X = coords[r * 4 + 1] >> 1, Y = coords[r * 4 + 2] >> 1, Z = coords[r * 4 + 3] >> 1;
        gridc[(long)b * Rc * Rc * Rc + ((long)X * Rc + Y) * Rc + Z] = 0;          // occupied (every writer stores the same value)
    }
}
// one work-group per cloud walks its Rc^3 cells in order: pass 0 counts the occupied ones, pass 1 numbers them from
// base[b] on (grid <- global row, VX_EMPTY stays) and writes their coords
__global__ __launch_bounds__(1024) void vx_number_kernel(int32_t* __restrict__ gridc, int Rc, const int32_t* __restrict__ base,
                                                          int32_t* __restrict__ counts, int32_t* __restrict__ coordsc, int pass) {
    __shared__ int wcnt[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int V = Rc * Rc * Rc;
    int32_t* g = gridc + (long)b * V;
    int run = pass ? base[b] : 0;
    for (int v0 = 0; v0 < V; v0 += 1024) {
        const int v = v0 + tid;
        const bool occ = v < V && g[v] != VX_EMPTY;
        const unsigned long long m = __ballot(occ);
        if (lane == 0) wcnt[wave] = __popcll(m);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int c = wcnt[w];
            before += (w < wave) ? c : 0;
            total += c;
        }
        if (pass && occ) {
            const int row = run + before + __popcll(m & ((1ull << lane) - 1ull));
            g[v] = row;
            coordsc[(long)row * 4] = b; coordsc[(long)row * 4 + 1] = v / (Rc * Rc);
            coordsc[(long)row * 4 + 2] = (v / Rc) % Rc; coordsc[(long)row * 4 + 3] = v % Rc;
        }
        run += total;
        __syncthreads();
    }
    if (!pass && tid == 0) counts[b] = run;
}
// child table of the coarse rows (8 slots, slot = (x&1)*4 + (y&1)*2 + (z&1), canonical fine rows only) and parent / slot
// of the fine rows
__global__ __launch_bounds__(256) void vx_child_kernel(const int32_t* __restrict__ coordsc, long rowsc, const int32_t* __restrict__ gridf,
                                                        int Rf, int32_t* __restrict__ child) {
    for (long e = (long)pm_xcd_contiguous(blockIdx.x, gridDim.x) * 256 + threadIdx.x; e < rowsc * 8; e += (long)gridDim.x * 256) {
        const long r = e >> 3;
        const int s = (int)(e & 7);
        const int b = coordsc[r * 4], X = 2 * coordsc[r * 4 + 1] + (s >> 2), Y = 2 * coordsc[r * 4 + 2] + ((s >> 1) & 1),
                  Z = 2 * coordsc[r * 4 + 3] + (s & 1);
        int v = -1;
        if (X < Rf && Y < Rf && Z < Rf) {
            const int g = gridf[(long)b * Rf * Rf * Rf + ((long)X * Rf + Y) * Rf + Z];
            v = g == VX_EMPTY ? -1 : g;
        }
        child[e] = v;
    }
}
__global__ __launch_bounds__(256) void vx_parent_kernel(const int32_t* __restrict__ coordsf, long rowsf, const int32_t* __restrict__ gridc,
                                                         int Rc, const int32_t* __restrict__ gridf, int Rf,
                                                         int32_t* __restrict__ parent, int32_t* __restrict__ parent_canon,
                                                         int32_t* __restrict__ slot) {
    for (long r = (long)pm_xcd_contiguous(blockIdx.x, gridDim.x) * 256 + threadIdx.x; r < rowsf; r += (long)gridDim.x * 256) {
        const int b = coordsf[r * 4], x = coordsf[r * 4 + 1], y = coordsf[r * 4 + 2], z = coordsf[r * 4 + 3];
        const int p = gridc[(long)b * Rc * Rc * Rc + ((long)(x >> 1) * Rc + (y >> 1)) * Rc + (z >> 1)];
        parent[r] = p;                                    // every row (a duplicate coordinate still reads its parent's features)
        // ... but only the canonical row of a coordinate is a child of that parent (it alone receives gradient from it)
        parent_canon[r] = gridf[(long)b * Rf * Rf * Rf + ((long)x * Rf + y) * Rf + z] == (int)r ? p : -1;
        slot[r] = (x & 1) * 4 + (y & 1) * 2 + (z & 1);
    }
}

// dst[r][j*C + c] = idx[r][j] >= 0 ? src[idx[r][j]][c] : 0        (C % 4 == 0: one float4 per thread)
__global__ __launch_bounds__(256) void rows_gather_kernel(const float* __restrict__ src, long lds, const int32_t* __restrict__ idx,
                                                           long rows, int J, int C, float* __restrict__ dst, long ldd) {
    const int c4 = C >> 2;
    const long total = rows * J * c4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long rj = e / c4;
        const int q = (int)(e - rj * c4);
        const long r = rj / J;
        const int j = (int)(rj - r * J);
        const int i = idx[rj];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i >= 0) v = *(const float4*)(src + (long)i * lds + 4 * q);
        *(float4*)(dst + r * ldd + (long)j * C + 4 * q) = v;
    }
}
// dsrc[r][c] (= or +=) (sum_j tidx[r][j] >= 0 ? dcols[tidx[r][j]][slot(r,j)*C + c] : 0) * (y ? 1 - y[r][c]^2 : 1)
// accumulate: 0 = overwrite, 1 = add the product to what dsrc holds, 2 = add what dsrc holds to the SUM, then multiply
// slot(r, j) = tslot ? tslot[r*J + j] : (mirror ? J - 1 - j ... see host) -- the host passes explicit conventions:
//   mode 0: block = j (table already arranged per block), mode 1: block = tslot[r*J + j], mode 2: block = 0 (plain rows)
__global__ __launch_bounds__(256) void rows_gather_bwd_kernel(const float* __restrict__ dcols, long ldc, const int32_t* __restrict__ tidx,
                                                               const int32_t* __restrict__ tslot, int mode, int reverse, int self_col,
                                                               long rows, int J, int C, const float* __restrict__ y, long ldy,
                                                               int accumulate, float* __restrict__ dsrc, long lds,
                                                               const int32_t* __restrict__ rowmap, const float* __restrict__ skip,
                                                               long ldskip, const int32_t* __restrict__ skipmap) {
    // skip / skipmap (nullable): a SPARSE raw contribution to add before the activation derivative -- row r receives
    // skip[skipmap[r]] when skipmap[r] >= 0 (the compact decoder backward: only the max-pool winners' ancestors carry a skip
    // gradient; the dense zero-filled tensor that accumulate == 2 reads back is never built)
    // rowmap (nullable): the table holds rows of a level, dcols only a SUBSET of them -- rowmap[row] = its row in dcols, or -1
    // (a compacted column gradient: network.py::_decoder_backward_compact)
    const int c4 = C >> 2;
    const long total = rows * c4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / c4;
        const int q = (int)(e - r * c4);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        // self_col >= 0: the row only has consumers if the table's own-cell entry points back at it (a duplicate coordinate
        // is never anybody's neighbour: its gradient is zero, not its canonical twin's)
        const bool live = self_col < 0 || tidx[r * J + self_col] == (int)r;
        if (live && J <= 8 && mode != 1) {
            // up to eight sources (the children of a coarse row): all indices first, then all rows -- eight independent loads in
            // flight instead of a chain of index -> row -> index -> row; same summation order
            int ix[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) ix[j] = j < J ? tidx[r * J + (reverse ? J - 1 - j : j)] : -1;
            if (rowmap) {
#pragma unroll
                for (int j = 0; j < 8; ++j) ix[j] = ix[j] >= 0 ? rowmap[ix[j]] : -1;
            }
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = ix[j] >= 0 ? *(const float4*)(dcols + (long)ix[j] * ldc + (long)(mode == 0 ? j : 0) * C + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (ix[j] >= 0) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
        } else {
            for (int j = 0; live && j < J; ++j) {
                int i = tidx[r * J + (reverse ? J - 1 - j : j)];
                if (i >= 0 && rowmap) i = rowmap[i];
                if (i >= 0) {
                    const int blk = mode == 0 ? j : (mode == 1 ? tslot[r * J + j] : 0);
                    const float4 v = *(const float4*)(dcols + (long)i * ldc + (long)blk * C + 4 * q);
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                }
            }
        }
        float4* o = (float4*)(dsrc + r * lds + 4 * q);
        if (accumulate == 2) {                             // dsrc holds a RAW contribution: (it + the gathered sum) * act'
            const float4 p = *o;
            s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
        }
        if (skipmap) {
            const int k = skipmap[r];
            if (k >= 0) {
                const float4 p = *(const float4*)(skip + (long)k * ldskip + 4 * q);
                s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
            }
        }
        if (y) {
            const float4 h = *(const float4*)(y + r * ldy + 4 * q);
            s.x *= 1.0f - h.x * h.x; s.y *= 1.0f - h.y * h.y; s.z *= 1.0f - h.z * h.z; s.w *= 1.0f - h.w * h.w;
        }
        if (accumulate == 1) {                             // dsrc holds a finished (pre-activation) contribution
            const float4 p = *o;
            s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
        }
        *o = s;
    }
}

// level-0 input features: [f, x / R, y / R, z / R] per row (4 columns)
__global__ __launch_bounds__(256) void vx_features0_kernel(const float* __restrict__ x, long ldx, int P, int C, int R, long rows,
                                                            const int32_t* __restrict__ coords, float* __restrict__ feat) {
    const float Rf = (float)R;                        // a true division per coordinate: bit-identical to the restatement's x / R
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
        const long b = r / P;
        const float f = C > 3 ? x[b * ldx + (r - b * P) * C + 3] : 1.0f;
        *(float4*)(feat + r * 4) = make_float4(f, __fdiv_rn((float)coords[r * 4 + 1], Rf), __fdiv_rn((float)coords[r * 4 + 2], Rf),
                                               __fdiv_rn((float)coords[r * 4 + 3], Rf));
    }
}

#define VX_LAUNCH(kern, n, ...) hipLaunchKernelGGL(kern, dim3(vx_blocks(n)), dim3(256), 0, pm_stream(stream), __VA_ARGS__)

extern "C" int pm_voxel_grid0_f32(const float* x, long ldx, int B, int P, int C, int R, int32_t* grid, int32_t* coords,
                                  float* feat, void* stream) {
    PM_REQUIRE(x && grid && coords && B > 0 && P > 0 && C >= 3 && R > 0 && R <= 256 && ldx >= (long)P * C);
    const long cells = (long)B * R * R * R, rows = (long)B * P;
    VX_LAUNCH(vx_fill_kernel, cells, grid, cells, VX_EMPTY);
    VX_LAUNCH(vx_grid0_kernel, rows, x, ldx, P, C, R, rows, grid, coords);
    if (feat) VX_LAUNCH(vx_features0_kernel, rows, x, ldx, P, C, R, rows, (const int32_t*)coords, feat);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

extern "C" int pm_voxel_nbr27_i32(const int32_t* coords, long rows, const int32_t* grid, int R, int32_t* nbr, int ld, void* stream) {
    PM_REQUIRE(coords && grid && nbr && rows > 0 && R > 0 && ld >= 27);
    VX_LAUNCH(vx_nbr27_kernel, rows * ld, coords, rows, grid, R, nbr, ld);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

extern "C" int pm_voxel_mirror27_i32(const int32_t* nbr, long rows, int32_t* out, void* stream) {
    PM_REQUIRE(nbr && out && rows > 0 && nbr != out);
    VX_LAUNCH(vx_mirror27_kernel, rows * 27, nbr, rows, out);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

extern "C" int pm_voxel_down_count_i32(const int32_t* coords_f, long rows_f, int B, int Rc, int32_t* grid_c, int32_t* counts,
                                       void* stream) {
    PM_REQUIRE(coords_f && grid_c && counts && rows_f > 0 && B > 0 && Rc > 0);
    const long cells = (long)B * Rc * Rc * Rc;
    VX_LAUNCH(vx_fill_kernel, cells, grid_c, cells, VX_EMPTY);
    VX_LAUNCH(vx_mark_kernel, rows_f, coords_f, rows_f, Rc, grid_c);
    hipLaunchKernelGGL(vx_number_kernel, dim3(B), dim3(1024), 0, pm_stream(stream), grid_c, Rc, (const int32_t*)nullptr, counts,
                       (int32_t*)nullptr, 0);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

extern "C" int pm_voxel_down_build_i32(const int32_t* coords_f, long rows_f, const int32_t* grid_f, int Rf, int B, int Rc,
                                       const int32_t* base, int32_t* grid_c, long rows_c, int32_t* coords_c, int32_t* child,
                                       int32_t* parent, int32_t* parent_canon, int32_t* slot, void* stream) {
    PM_REQUIRE(coords_f && grid_f && base && grid_c && coords_c && child && parent && parent_canon && slot && rows_f > 0 &&
               rows_c > 0 && B > 0);
    hipLaunchKernelGGL(vx_number_kernel, dim3(B), dim3(1024), 0, pm_stream(stream), grid_c, Rc, base, (int32_t*)nullptr, coords_c, 1);
    VX_LAUNCH(vx_child_kernel, rows_c * 8, (const int32_t*)coords_c, rows_c, grid_f, Rf, child);
    VX_LAUNCH(vx_parent_kernel, rows_f, coords_f, rows_f, (const int32_t*)grid_c, Rc, grid_f, Rf, parent, parent_canon, slot);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

extern "C" int pm_rows_gather_f32(const float* src, long lds, const int32_t* idx, long rows, int J, int C, float* dst, long ldd,
                                  void* stream) {
    PM_REQUIRE(src && idx && dst && rows > 0 && J > 0 && C > 0 && C % 4 == 0 && lds >= C && ldd >= (long)J * C);
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) != 0 || lds % 4 != 0 || ldd % 4 != 0) return PM_EALIGN;
    VX_LAUNCH(rows_gather_kernel, rows * J * (C / 4), src, lds, idx, rows, J, C, dst, ldd);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

extern "C" int pm_rows_gather_bwd_mapped_f32(const float* dcols, long ldc, const int32_t* tidx, const int32_t* tslot, int mode,
                                             int reverse, int self_col, long rows, int J, int C, const float* y_tanh, long ldy,
                                             int accumulate, float* dsrc, long lds, const int32_t* rowmap, void* stream) {
    PM_REQUIRE(dcols && tidx && dsrc && rows > 0 && J > 0 && C > 0 && C % 4 == 0 && mode >= 0 && mode <= 2 && (mode != 1 || tslot) &&
               self_col < J);
    if ((((uintptr_t)dcols | (uintptr_t)dsrc | (uintptr_t)y_tanh) & 15) != 0 || ldc % 4 != 0 || lds % 4 != 0 || (y_tanh && ldy % 4 != 0))
        return PM_EALIGN;
    VX_LAUNCH(rows_gather_bwd_kernel, rows * (C / 4), dcols, ldc, tidx, tslot, mode, reverse, self_col, rows, J, C, y_tanh, ldy,
              accumulate, dsrc, lds, rowmap, (const float*)nullptr, 0L, (const int32_t*)nullptr);
    PM_CHECK_LAUNCH();
    return PM_OK;
}
extern "C" int pm_rows_gather_bwd_skip_f32(const float* dcols, long ldc, const int32_t* tidx, const int32_t* tslot, int mode,
                                           int reverse, int self_col, long rows, int J, int C, const float* y_tanh, long ldy,
                                           float* dsrc, long lds, const float* skip, long ldskip, const int32_t* skipmap,
                                           void* stream) {
    PM_REQUIRE(dcols && tidx && dsrc && skip && skipmap && rows > 0 && J > 0 && C > 0 && C % 4 == 0 && mode >= 0 && mode <= 2 &&
               (mode != 1 || tslot) && self_col < J && ldskip >= C);
    if ((((uintptr_t)dcols | (uintptr_t)dsrc | (uintptr_t)y_tanh | (uintptr_t)skip) & 15) != 0 || ldc % 4 != 0 || lds % 4 != 0 ||
        ldskip % 4 != 0 || (y_tanh && ldy % 4 != 0))
        return PM_EALIGN;
    VX_LAUNCH(rows_gather_bwd_kernel, rows * (C / 4), dcols, ldc, tidx, tslot, mode, reverse, self_col, rows, J, C, y_tanh, ldy, 0,
              dsrc, lds, (const int32_t*)nullptr, skip, ldskip, skipmap);
    PM_CHECK_LAUNCH();
    return PM_OK;
}
extern "C" int pm_rows_gather_bwd_f32(const float* dcols, long ldc, const int32_t* tidx, const int32_t* tslot, int mode, int reverse,
                                      int self_col, long rows, int J, int C, const float* y_tanh, long ldy, int accumulate,
                                      float* dsrc, long lds, void* stream) {
    return pm_rows_gather_bwd_mapped_f32(dcols, ldc, tidx, tslot, mode, reverse, self_col, rows, J, C, y_tanh, ldy, accumulate, dsrc, lds,
                                         nullptr, stream);
}


// ---------------------------------------------------------------------------------------------------------------------------
// Row bookkeeping of the compact decoder backward (network.py::SparseUNet._decoder_backward_compact) -- the cloud-wide max-pool
// leaves at most S = c0 rows per cloud with a gradient; these kernels name those rows and their ancestors per level, sum the
// children of a coarse row and build row -> compact-slot maps, all in fixed order (no sort library, no atomics on data).

// One wave per cloud, S <= 64 ids: v[i] = src[b][i] (+ b * row_base), through `map` (v == pad_in -> pad_out); u[b] = the
// cloud's DISTINCT mapped ids ascending, padded with pad_out; um = the same with -1 in the padding slots; rank[b][i] = the slot
// of id i in u[b].  All-pairs through the wave (S shuffles): rank = number of distinct smaller ids.
__global__ __launch_bounds__(64) void rows_uniq_kernel(const int32_t* __restrict__ src, long lds, int B, int S, long row_base,
                                                        const int32_t* __restrict__ map, int pad_in, int pad_out,
                                                        int32_t* __restrict__ u, int32_t* __restrict__ um,
                                                        int32_t* __restrict__ rank) {
    const int b = blockIdx.x, i = threadIdx.x;
    int v = 0x7fffffff;
    if (i < S) {
        const int raw = src[(long)b * lds + i];
        if (map) v = raw == pad_in ? pad_out : map[raw];
        else v = (int)(raw + (long)b * row_base);
    }
    // first occurrence of its value? (lanes >= S hold INT_MAX and are ignored)
    bool first = i < S;
    for (int j = 0; j < S; ++j) {
        const int vj = __shfl(v, j, 64);
        if (j < i && vj == v) first = false;
    }
    int less = 0;
    for (int j = 0; j < S; ++j) {
        const int vj = __shfl(v, j, 64);
        const bool fj = __shfl((int)first, j, 64) != 0;
        if (fj && vj < v) ++less;
    }
    const unsigned long long firsts = __ballot(first && v != pad_out);
    const int ndist = __popcll(firsts);                      // distinct live ids (the pad id sorts last: pad_out >= every row id)
    if (i < S) {
        rank[(long)b * S + i] = less;
        if (first && v != pad_out) {
            u[(long)b * S + less] = v;
            um[(long)b * S + less] = v;
        }
        if (i >= ndist) {
            u[(long)b * S + i] = pad_out;
            um[(long)b * S + i] = -1;
        }
    }
}

extern "C" int pm_rows_uniq_i32(const int32_t* src, long lds, int B, int S, long row_base, const int32_t* map, int pad_in,
                                int pad_out, int32_t* u, int32_t* um, int32_t* rank, void* stream) {
    PM_REQUIRE(src && u && um && rank && B > 0 && S > 0 && S <= 64 && lds >= S && pad_out >= 0);
    hipLaunchKernelGGL(rows_uniq_kernel, dim3(B), dim3(64), 0, pm_stream(stream), src, lds, B, S, row_base, map, pad_in, pad_out, u, um,
                       rank);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// y[b*S + j][:] = sum over i (ascending) with rank[b][i] == j of x[b*S + i][:]   -- the children of compact coarse row j
__global__ __launch_bounds__(256) void child_sum_kernel(const float* __restrict__ x, long ldx, const int32_t* __restrict__ rank,
                                                         int S, int C, float* __restrict__ y, long ldy) {
    __shared__ int rk[64];
    const long b = blockIdx.x;
    if (threadIdx.x < S) rk[threadIdx.x] = rank[b * S + threadIdx.x];
    __syncthreads();
    const int c4n = C >> 2;
    for (int e = threadIdx.x; e < S * c4n; e += 256) {
        const int j = e / c4n, q = e - j * c4n;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = 0; i < S; ++i)
            if (rk[i] == j) {
                const float4 v = *(const float4*)(x + (b * S + i) * ldx + 4 * q);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        *(float4*)(y + (b * S + j) * ldy + 4 * q) = s;
    }
}

extern "C" int pm_child_sum_f32(const float* x, long ldx, const int32_t* rank, int B, int S, int C, float* y, long ldy, void* stream) {
    PM_REQUIRE(x && rank && y && B > 0 && S > 0 && S <= 64 && C > 0 && C % 4 == 0 && ldx >= C && ldy >= C);
    if ((((uintptr_t)x | (uintptr_t)y) & 15) != 0 || ldx % 4 != 0 || ldy % 4 != 0) return PM_EALIGN;
    hipLaunchKernelGGL(child_sum_kernel, dim3(B), dim3(256), 0, pm_stream(stream), x, ldx, rank, S, C, y, ldy);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// map[0 .. n) = -1, then map[ids[k]] = k for the ids that are not `pad` (ids distinct apart from the padding)
__global__ __launch_bounds__(256) void rowmap_scatter_kernel(int32_t* __restrict__ map, long n, const int32_t* __restrict__ ids, long N,
                                                              int pad) {
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < N; k += (long)gridDim.x * 256) {
        const int r = ids[k];
        if (r != pad && r >= 0 && r < n) map[r] = (int)k;
    }
}

extern "C" int pm_rowmap_scatter_i32(int32_t* map, long n, const int32_t* ids, long N, int pad, void* stream) {
    PM_REQUIRE(map && ids && n > 0 && N > 0);
    if (hipMemsetAsync(map, 0xFF, (size_t)n * sizeof(int32_t), pm_stream(stream)) != hipSuccess) return PM_EINVAL;
    VX_LAUNCH(rowmap_scatter_kernel, N, map, n, ids, N, pad);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// out[k][:] = sel[k] >= 0 ? table[sel[k]][:] : -1      (rows of an index table picked by row id; -1 rows for the padding)
__global__ __launch_bounds__(256) void table_rows_kernel(const int32_t* __restrict__ table, long ldt, int J, const int32_t* __restrict__ sel,
                                                          long N, int32_t* __restrict__ out) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < N * J; e += (long)gridDim.x * 256) {
        const long k = e / J;
        const int j = (int)(e - k * J);
        const int r = sel[k];
        out[e] = r >= 0 ? table[(long)r * ldt + j] : -1;
    }
}

extern "C" int pm_table_rows_i32(const int32_t* table, long ldt, int J, const int32_t* sel, long N, int32_t* out, void* stream) {
    PM_REQUIRE(table && sel && out && J > 0 && ldt >= J && N > 0);
    VX_LAUNCH(table_rows_kernel, N * J, table, ldt, J, sel, N, out);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// (rows, m + 1) gather table of a virtual [unpool | skip] operand: m chunks of the parent's row, then the row's own behind them
__global__ __launch_bounds__(256) void vcat_table_kernel(const int32_t* __restrict__ parent, long rows, int m, long rows_hi,
                                                          int32_t* __restrict__ out) {
    const int J = m + 1;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < rows * J; e += (long)gridDim.x * 256) {
        const long r = e / J;
        const int j = (int)(e - r * J);
        out[e] = j < m ? parent[r] * m + j : (int)(r + (long)m * rows_hi);
    }
}

extern "C" int pm_voxel_vcat_table_i32(const int32_t* parent, long rows, int m, long rows_hi, int32_t* out, void* stream) {
    PM_REQUIRE(parent && out && rows > 0 && m > 0 && rows_hi > 0 && rows + (long)m * rows_hi < (1L << 31));
    VX_LAUNCH(vcat_table_kernel, rows * (m + 1), parent, rows, m, rows_hi, out);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// base[i] = counts[0] + ... + counts[i-1], total[0] = the sum (one work-group; n clouds)
__global__ __launch_bounds__(1024) void exscan_kernel(const int32_t* __restrict__ counts, int n, int32_t* __restrict__ base,
                                                       int32_t* __restrict__ total) {
    __shared__ int sr[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int lo = 0; lo < n; lo += 1024) {
        const int i = lo + threadIdx.x;
        const int c = i < n ? counts[i] : 0;
        sr[threadIdx.x] = c;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int a = threadIdx.x >= o ? sr[threadIdx.x - o] : 0;
            __syncthreads();
            sr[threadIdx.x] += a;
            __syncthreads();
        }
        if (i < n) base[i] = carry + sr[threadIdx.x] - c;
        __syncthreads();
        if (threadIdx.x == 1023) carry += sr[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) total[0] = carry;
}

extern "C" int pm_exclusive_scan_i32(const int32_t* counts, int n, int32_t* base, int32_t* total, void* stream) {
    PM_REQUIRE(counts && base && total && n > 0);
    hipLaunchKernelGGL(exscan_kernel, dim3(1), dim3(1024), 0, pm_stream(stream), counts, n, base, total);
    PM_CHECK_LAUNCH();
    return PM_OK;
}
